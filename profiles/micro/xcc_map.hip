// Which XCD does workgroup b of a grid land on?  (blockIdx % 8 == XCC_ID is observed, not guaranteed.)
// hipcc --offload-arch=gfx950 -O2 -o xcc_map xcc_map.hip && ./xcc_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(uint32_t* out, int spin)
{
    extern __shared__ uint32_t lds[];
    const uint32_t x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
    lds[threadIdx.x] = x;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = 0; i < spin; i++) acc += lds[(threadIdx.x + i) & 255];
    if (threadIdx.x == 0) out[blockIdx.x] = x | (acc & 0x80000000u);
}
int main()
{
    for (int cfg = 0; cfg < 4; cfg++) {
        const int grid = cfg == 0 ? 256 : (cfg == 1 ? 1024 : 2048), nt = cfg == 3 ? 1024 : 512;
        const size_t lds = cfg == 3 ? 120 * 1024 : 78 * 1024;
        uint32_t* d; hipMalloc(&d, grid * 4);
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(grid), dim3(nt), lds, 0, d, 20000);
        std::vector<uint32_t> h(grid);
        hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
        int match = 0, hist[8][8] = {};
        for (int b = 0; b < grid; b++) { const int x = h[b] & 15; hist[b % 8][x & 7]++; }
        // best permutation-free check: is the map b%8 -> xcc a function?
        int func = 0;
        for (int r = 0; r < 8; r++) { int mx = 0; for (int c = 0; c < 8; c++) mx = hist[r][c] > mx ? hist[r][c] : mx; func += mx; }
        printf("grid %d x %d threads, lds %zu: %d of %d blocks follow a fixed (blockIdx %% 8 -> XCC) map\n", grid, nt, lds, func, grid);
        for (int r = 0; r < 8; r++) { printf("  b%%8=%d:", r); for (int c = 0; c < 8; c++) printf(" %4d", hist[r][c]); printf("\n"); }
        (void)match;
        hipFree(d);
    }
    return 0;
}
