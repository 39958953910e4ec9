// Round trip of a small result to the host after a kernel: hipMemcpyAsync into pageable memory (what the pipeline did through round 4)
// against a pinned staging buffer, and against a kernel that writes into mapped pinned memory itself (no copy at all).
// hipcc --offload-arch=gfx950 -O3 -o readback readback.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_bump(unsigned long long* p, unsigned long long* mapped) { if (threadIdx.x == 0) { p[0] += 1; if (mapped) { mapped[0] = p[0]; __threadfence_system(); } } }
__global__ void k_fetch4(const unsigned* __restrict__ host, unsigned* __restrict__ dev, int w0, int w1, int w2, int w3)
{
    const int w[4] = {w0, w1, w2, w3};
    for (int r = 0; r < 4; r++) for (int i = threadIdx.x; i < w[r]; i += 256) dev[r * 2048 + i] = host[r * 2048 + i];
}
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
    // no hipStreamSynchronize at all: the kernel's last store is a sequence number in mapped memory, the host spins on it
    {
        volatile unsigned long long* flag = mapped;
        unsigned long long base; CK(hipMemcpy(&base, d, 8, hipMemcpyDeviceToHost));
        for (int chain : {1, 4}) {                         // (chain: launches queued before the one that publishes)
            const double t1 = now();
            for (int r = 0; r < R; r++) {
                for (int c = 1; c < chain; c++) hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d + 8, (unsigned long long*)nullptr);
                hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, dmapped);
                base++;
                while (*flag != base) { }
            }
            printf("kernel writes mapped memory, host spins on it (%d launches per trip)  %6.1f us\n", chain, (now() - t1) / R * 1e6);
            const double t2 = now();
            for (int r = 0; r < R; r++) {
                for (int c = 1; c < chain; c++) hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d + 8, (unsigned long long*)nullptr);
                hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, (unsigned long long*)nullptr);
                CK(hipMemcpyAsync(pin, d, 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                base++;
            }
            printf("pinned staging + synchronize              (%d launches per trip)  %6.1f us\n", chain, (now() - t2) / R * 1e6);
        }
    }
    // the other direction: four small tables for the next kernel (code table 512 B, digit map 5 KB, character model 36 B, digit
    // values 6 KB) as four hipMemcpyAsync from pageable memory, against one memcpy each into mapped pinned memory + ONE kernel
    // that fetches them (k_fetch)
    {
        const int sz[4] = {512, 5120, 64, 6144};
        unsigned char* dd; CK(hipMalloc(&dd, 65536));
        unsigned char *up, *dup; CK(hipHostMalloc(&up, 65536, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&dup, up, 0));
        static unsigned char src[4][8192];
        for (int mode = 0; mode < 2; mode++) {
            CK(hipStreamSynchronize(st));
            const double t1 = now();
            double host = 0;
            for (int r = 0; r < R; r++) {
                const double h0 = now();
                int off = 0;
                for (int i = 0; i < 4; i++) {
                    if (mode == 0) CK(hipMemcpyAsync(dd + off, src[i], sz[i], hipMemcpyHostToDevice, st));
                    else memcpy(up + off, src[i], sz[i]);
                    off += 8192;
                }
                if (mode == 1) hipLaunchKernelGGL(k_fetch4, dim3(1), dim3(256), 0, st, (const unsigned*)dup, (unsigned*)dd, sz[0] / 4, sz[1] / 4, sz[2] / 4, sz[3] / 4);
                hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, (unsigned long long*)nullptr);
                host += now() - h0;
                CK(hipStreamSynchronize(st));
            }
            printf("four small tables to the device + kernel: %-34s %6.1f us per trip, %5.1f us of it on the host before the kernel is queued\n",
                   mode == 0 ? "4 x hipMemcpyAsync (pageable)" : "mapped staging + one fetch kernel", (now() - t1) / R * 1e6, host / R * 1e6);
        }
    }
    // the kernel alone
    const double t0 = now();
    for (int r = 0; r < R; r++) { hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d, (unsigned long long*)nullptr); CK(hipStreamSynchronize(st)); }
    printf("kernel + synchronize alone              %6.1f us\n", (now() - t0) / R * 1e6);
    return 0;
}
