// level2_pattern.hip -- the memory side of k_msd_scatter's level 2 ALONE, for two output record sizes (12 and 8 bytes):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o level2_pattern level2_pattern.hip && ./level2_pattern
// 1.5 G records in segments of SEG records (the first-digit buckets: 1.1 M on the 3.1 Gb stand-in); a tile of NB x r records is READ
// front to back as 12-byte records and its records written to the NB sub-buckets of the tile's segment, r per sub-bucket, slots
// claimed on the sub-bucket's cursor with one returning atomic per (tile, sub-bucket); the workgroups of group g = blockIdx % 8 (one
// XCD) work on the same segment together, as the tile list of k_msd_scatter deals them.  No ranking, no LDS staging: what the
// pattern costs whatever the compute side does.  Question (round 5): what does an 8-byte record (32-bit key remainder + index)
// at the OUTPUT of level 2 buy?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec12 { uint32_t a, b, c; };
struct Rec8 { uint32_t a, b; };

template <typename OUT>
__global__ void __launch_bounds__(512)
k_level2(const Rec12* __restrict__ in, OUT* __restrict__ out, uint32_t* __restrict__ cursor, uint32_t NB, uint32_t r, uint32_t SEG,
         uint32_t nseg)
{
    extern __shared__ uint32_t s_base[];
    const uint32_t g = blockIdx.x & 7u, w = blockIdx.x >> 3, R = gridDim.x >> 3;
    const uint32_t tile = NB * r, tps = SEG / tile;           // tiles per segment
    for (uint32_t seg = g; seg < nseg; seg += 8) {
        for (uint32_t t = w; t < tps; t += R) {
            __syncthreads();
            for (uint32_t d = threadIdx.x; d < NB; d += 512)
                s_base[d] = seg * SEG + d * (SEG / NB) + atomicAdd(&cursor[(size_t)seg * NB + d], r);
            __syncthreads();
            const Rec12* src = in + (size_t)seg * SEG + (size_t)t * tile;
            for (uint32_t j = threadIdx.x; j < tile; j += 512) {
                const Rec12 v = src[j];
                const uint32_t d = j / r, o = j - d * r;
                if constexpr (sizeof(OUT) == 12) out[(size_t)s_base[d] + o] = OUT{v.a, v.b, v.c};
                else out[(size_t)s_base[d] + o] = OUT{v.a ^ v.b, v.c};
            }
        }
    }
}

int main(int argc, char** argv)
{
    const uint32_t NB = argc > 1 ? (uint32_t)atoi(argv[1]) : 1400u;
    const uint32_t nseg = 1364;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t r : {1u, 2u, 3u, 6u}) {
        const uint32_t tile = NB * r;
        const uint32_t SEG = (1100000u / tile) * tile;
        const uint64_t records = (uint64_t)SEG * nseg;
        Rec12* in; void* out; uint32_t* cursor;
        CK(hipMalloc(&in, records * 12)); CK(hipMalloc(&out, records * 12)); CK(hipMalloc(&cursor, (size_t)nseg * NB * 4));
        CK(hipMemset(in, 1, records * 12));
        for (int sz : {12, 8}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(cursor, 0, (size_t)nseg * NB * 4));
                CK(hipEventRecord(e0));
                if (sz == 12) hipLaunchKernelGGL(k_level2<Rec12>, dim3(1024), dim3(512), (size_t)NB * 4, 0, in, (Rec12*)out, cursor, NB, r, SEG, nseg);
                else hipLaunchKernelGGL(k_level2<Rec8>, dim3(1024), dim3(512), (size_t)NB * 4, 0, in, (Rec8*)out, cursor, NB, r, SEG, nseg);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            printf("sub-buckets %5u  run %u rec  out %2d B/rec : %7.2f ms  (%.3g records, %.1f GB read, %.1f GB written)\n", NB, r, sz, best,
                   (double)records, records * 12.0 / 1e9, records * (double)sz / 1e9);
        }
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(cursor));
    }
    return 0;
}
