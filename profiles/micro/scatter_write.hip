// scatter_write.hip -- what the memory side takes of the radix-partition kernel's STORE PATTERN alone (no text, no LDS ranking):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scatter_write scatter_write.hip && ./scatter_write [records] [bins]
// A tile is bins x r records of 12 bytes; the workgroup claims r slots per (group, bin) cursor (group = blockIdx % 8, one returning
// atomic per bin and tile, as k_msd_part_text does) and consecutive lanes store consecutive records of a bin's run.  The sweep over
// r (3 ... 192 records = 36 ... 2 304 bytes per run) shows what run length buys; r = 12 with 2 700 bins is the pattern of the
// 3.1 Gb stand-in at N = 1 (31 700 records of a 65 536-position tile over ~2 700 first digits).  "stream": the same bytes
// written front to back by the same grid -- the box's plain write rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec { uint32_t a, b, c; };

__global__ void __launch_bounds__(1024)
k_scatter(Rec* __restrict__ out, const uint32_t* __restrict__ bucket_base, uint32_t* __restrict__ cursor, uint32_t NB, uint32_t r,
          uint32_t tiles)
{
    extern __shared__ uint32_t s_base[];
    const uint32_t g = blockIdx.x & 7u;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < NB; d += 1024) s_base[d] = bucket_base[d] + atomicAdd(&cursor[(size_t)g * NB + d], r);
        __syncthreads();
        const uint32_t total = NB * r;
        for (uint32_t j = threadIdx.x; j < total; j += 1024) {
            const uint32_t d = j / r, o = j - d * r;
            out[(size_t)s_base[d] + o] = Rec{j, t, d};
        }
    }
}

__global__ void __launch_bounds__(1024)
k_stream(Rec* __restrict__ out, uint64_t records)
{
    for (uint64_t j = (uint64_t)blockIdx.x * 1024 + threadIdx.x; j < records; j += (uint64_t)gridDim.x * 1024)
        out[j] = Rec{(uint32_t)j, 1u, 2u};
}

int main(int argc, char** argv)
{
    const uint64_t records = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1500000000ull;
    const uint32_t NB = argc > 2 ? (uint32_t)atoi(argv[2]) : 2700u;
    Rec* out; uint32_t* cursor; uint32_t* base;
    CK(hipMalloc(&out, (records + (uint64_t)NB * 4096) * sizeof(Rec)));
    CK(hipMalloc(&cursor, (size_t)8 * NB * 4));
    CK(hipMalloc(&base, (size_t)NB * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t grid = 1280;
    {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_stream, dim3(grid), dim3(1024), 0, 0, out, records); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("stream                      : %7.2f ms  %6.0f GB/s\n", best, records * 12.0 / best / 1e6);
    }
    for (uint32_t r : {3u, 6u, 12u, 24u, 48u, 96u, 192u}) {
        const uint32_t tiles = (uint32_t)(records / ((uint64_t)NB * r));
        const uint64_t written = (uint64_t)tiles * NB * r;
        // a bin holds tiles * r records, the 8 groups' claims interleaved inside it
        std::vector<uint32_t> hb(NB);
        for (uint32_t d = 0; d < NB; d++) hb[d] = (uint32_t)((uint64_t)d * tiles * r);
        CK(hipMemcpy(base, hb.data(), (size_t)NB * 4, hipMemcpyHostToDevice));
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            // every group claims from ONE cursor per bin in the real kernel only within its group; here the 8 groups share the bin's
            // range: group g's claims start at g * (its share) -- set the cursors so that the groups' runs interleave per tile
            std::vector<uint32_t> hc((size_t)8 * NB, 0u);
            const uint32_t per_group = (tiles + 7) / 8 * r;
            for (uint32_t g = 0; g < 8; g++) for (uint32_t d = 0; d < NB; d++) hc[(size_t)g * NB + d] = g * per_group;
            CK(hipMemcpy(cursor, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(1024), (size_t)NB * 4, 0, out, base, cursor, NB, r, tiles);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double runs = (double)tiles * NB, lines = written * 12.0 / 64.0 + runs * (1.0 - 12.0 / 64.0) * 0.0;   // (lower bound: payload / 64)
        printf("bins %5u  run %4u rec (%5u B): %7.2f ms  %6.0f GB/s  >= %5.1f G lines/s  (%.3g runs)\n", NB, r, r * 12, best,
               written * 12.0 / best / 1e6, lines / best / 1e6, runs);
    }
    return 0;
}
