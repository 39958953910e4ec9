// load_width.hip -- the LOAD phase of the leaf sort alone: 1.5 G records read window by window (4 096 records, 512 threads, 8 records
// per thread, slot e * 512 + thread, two workgroups per CU, the next window's loads issued before the current one is "used") as
// 12-byte records (dwordx3), 8-byte records (dwordx2) and 8-byte records two at a time (dwordx4):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o load_width load_width.hip && ./load_width
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct R12 { uint32_t a, b, c; };
struct R8 { uint32_t a, b; };

template <int MODE>
__global__ void __launch_bounds__(512)
k_load(const void* __restrict__ in, uint64_t records, uint32_t* __restrict__ sink)
{
    const uint64_t W = records / 4096;
    uint32_t acc = 0;
    for (uint64_t w = blockIdx.x; w < W; w += gridDim.x) {
        const uint64_t base = w * 4096;
        if (MODE == 12) {
            const R12* p = (const R12*)in + base;
#pragma unroll
            for (int e = 0; e < 8; e++) { const R12 r = p[e * 512 + threadIdx.x]; acc += r.a ^ r.b ^ r.c; }
        } else if (MODE == 8) {
            const R8* p = (const R8*)in + base;
#pragma unroll
            for (int e = 0; e < 8; e++) { const R8 r = p[e * 512 + threadIdx.x]; acc += r.a ^ r.b; }
        } else {
            const uint4* p = (const uint4*)((const R8*)in + base);
#pragma unroll
            for (int e = 0; e < 4; e++) { const uint4 r = p[e * 512 + threadIdx.x]; acc += r.x ^ r.y ^ r.z ^ r.w; }
        }
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

int main()
{
    const uint64_t records = 1500000000ull / 4096 * 4096;
    void* buf; uint32_t* sink;
    CK(hipMalloc(&buf, records * 12)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(buf, 1, records * 12));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode : {12, 8, 16}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 12) hipLaunchKernelGGL(k_load<12>, dim3(2048), dim3(512), 0, 0, buf, records, sink);
            else if (mode == 8) hipLaunchKernelGGL(k_load<8>, dim3(2048), dim3(512), 0, 0, buf, records, sink);
            else hipLaunchKernelGGL(k_load<16>, dim3(2048), dim3(512), 0, 0, buf, records, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double bytes = records * (mode == 12 ? 12.0 : 8.0);
        printf("%s: %6.2f ms  %5.0f GB/s  (%.1f GB)\n", mode == 12 ? "12-byte records, dwordx3" : mode == 8 ? " 8-byte records, dwordx2" : " 8-byte records, dwordx4 (two per load)", best, bytes / best / 1e6, bytes / 1e9);
    }
    return 0;
}
