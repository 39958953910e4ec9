// Round trip of a small result to the host after a kernel: hipMemcpyAsync into pageable memory (what the pipeline did through round 4)
// against a pinned staging buffer, and against a kernel that writes into mapped pinned memory itself (no copy at all).
// hipcc --offload-arch=gfx950 -O3 -o readback readback.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_bump(unsigned long long* p, unsigned long long* mapped) { if (threadIdx.x == 0) { p[0] += 1; if (mapped) { mapped[0] = p[0]; __threadfence_system(); } } }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *d, *pin, *mapped, *dmapped; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    CK(hipHostMalloc(&pin, 4096, hipHostMallocDefault));
    CK(hipHostMalloc(&mapped, 4096, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&dmapped, mapped, 0));
    unsigned long long pageable[16];
    const int R = 2000;
    for (int bytes : {8, 128, 4096}) {
        for (int mode = 0; mode < 3; mode++) {
            for (int w = 0; w < 50; w++) { hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, (unsigned long long*)nullptr); CK(hipStreamSynchronize(st)); }
            const double t0 = now();
            for (int r = 0; r < R; r++) {
                hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, mode == 2 ? dmapped : (unsigned long long*)nullptr);
                if (mode == 0) CK(hipMemcpyAsync(bytes <= 128 ? (void*)pageable : (void*)(new char[4096]), d, bytes, hipMemcpyDeviceToHost, st));
                if (mode == 1) CK(hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            }
            printf("%5d bytes  %-28s %6.1f us per kernel + read back\n", bytes, mode == 0 ? "pageable destination" : mode == 1 ? "pinned staging" : "kernel writes mapped memory", (now() - t0) / R * 1e6);
        }
    }
    // the kernel alone
    const double t0 = now();
    for (int r = 0; r < R; r++) { hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, (unsigned long long*)nullptr); CK(hipStreamSynchronize(st)); }
    printf("kernel + synchronize alone              %6.1f us\n", (now() - t0) / R * 1e6);
    return 0;
}
