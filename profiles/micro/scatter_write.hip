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
#include <string.h>
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

// The direct pattern with the workgroups of an XCD IN LOCKSTEP (round 5, a hypothesis): every tile's stores sweep the bins in
// order; if the 64 workgroups of a group (= one XCD) start their tiles together, a bin's 64 adjacent runs (64 x r records: the
// claims on a (group, bin) cursor are adjacent) land in that XCD's L2 within a short window, complete whole kilobytes there and
// leave for HBM together -- long runs without longer tiles.  A barrier per tile among the co-resident workgroups of a group:
// one atomic counter per group (a line of its own), spinning with s_sleep.  The grid must be resident as a whole (512 workgroups
// of 1 024 threads = two per CU).
__global__ void __launch_bounds__(1024)
k_scatter_lockstep(Rec* __restrict__ out, const uint32_t* __restrict__ bucket_base, uint32_t* __restrict__ cursor, uint32_t NB, uint32_t r,
                   uint32_t tiles, uint32_t* __restrict__ gbar, uint32_t nsync)
{
    // nsync = barriers per tile: 0 = none (the control: the same persistent grid, unsynchronised), 1 = one before the stores,
    // k > 1 = one before each k-th of the bin sweep
    extern __shared__ uint32_t s_base[];
    const uint32_t g = blockIdx.x & 7u, per_group = gridDim.x >> 3;
    uint32_t epoch = 0;
    const uint32_t iters = (tiles + gridDim.x - 1) / gridDim.x;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t t = it * gridDim.x + blockIdx.x;
        __syncthreads();
        if (t < tiles)
            for (uint32_t d = threadIdx.x; d < NB; d += 1024) s_base[d] = bucket_base[d] + atomicAdd(&cursor[(size_t)g * NB + d], r);
        __syncthreads();
        const uint32_t total = NB * r, parts = nsync ? nsync : 1u;
        for (uint32_t q = 0; q < parts; q++) {
            if (nsync) {
                epoch++;
                if (threadIdx.x == 0) {              // the group's workgroups start (this part of) their stores together
                    atomicAdd(&gbar[g * 64u], 1u);
                    while (__hip_atomic_load(&gbar[g * 64u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * per_group) __builtin_amdgcn_s_sleep(2);
                }
                __syncthreads();
            }
            if (t < tiles) {
                const uint32_t lo = (uint32_t)((uint64_t)total * q / parts), hi = (uint32_t)((uint64_t)total * (q + 1) / parts);
                for (uint32_t j = lo + threadIdx.x; j < hi; j += 1024) {
                    const uint32_t d = j / r, o = j - d * r;
                    out[(size_t)s_base[d] + o] = Rec{j, t, d};
                }
            }
        }
    }
}

// The reverse walk: one resident workgroup per (XCD, bin-group) pulls the bin's records from tile-ordered staging (each tile's
// run of r records sits at tile * NB * r + d * r, i.e. the reads are the scattered side) and writes them contiguously.  Reads are
// sector-granular and clean, so if the asymmetry between scattered reads and scattered writes is large this ordering wins.
__global__ void __launch_bounds__(1024)
k_gather_runs(Rec* __restrict__ out, const Rec* __restrict__ in, uint32_t NB, uint32_t r, uint32_t tiles)
{
    // workgroup b owns bins b, b + grid, ...; wave w takes tiles w, w + 16, ...; a wave reads r records of a tile's run per step
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t d = blockIdx.x; d < NB; d += gridDim.x) {
        const size_t obase = (size_t)d * tiles * r;
        // 64 lanes cover floor(64 / r) runs at once
        const uint32_t per = 64u / r, sub = lane / r, o = lane - sub * r;
        for (uint32_t t0 = wave * per; t0 < tiles; t0 += 16u * per) {
            const uint32_t t = t0 + sub;
            if (sub < per && t < tiles) {
                const Rec v = in[(size_t)t * NB * r + (size_t)d * r + o];
                out[obase + (size_t)t * r + o] = v;
            }
        }
    }
}

// XCD-local write combining (the micro-benchmark).  Per (group = XCD, bin) a staging block of
// BLK records in global memory that is only ever touched by that XCD (so it lives in that XCD's L2: 2 700 x 1 152 B = 3.1 MB of
// 4 MB); a tile APPENDS its r-record run to the block (returning atomic on the staging cursor, wrap-around inside the block) and
// the workgroup whose append crosses the end of the block FLUSHES it: reads the BLK records back (L2 hits, if the block stayed)
// and writes them as ONE run of BLK records to the bin's place in the output.  This is the TRAFFIC of the scheme and nothing
// else -- no completion counters, nobody waits for the other appenders of a block, the flushed bytes may be stale --: a LOWER
// bound on what a correct implementation would take.  nt: the final stores carry the non-temporal hint (they should not push
// the staging blocks out of the L2).
template <bool NT_STORE>
__global__ void __launch_bounds__(1024)
k_combine(Rec* __restrict__ out, Rec* __restrict__ stage, const uint32_t* __restrict__ bucket_base, uint32_t* __restrict__ cursor,
          uint32_t* __restrict__ scur, uint32_t NB, uint32_t r, uint32_t BLK, uint32_t tiles)
{
    extern __shared__ uint32_t s_base[];           // NB staging offsets, then the flush list (bin, final base) x NB, then its length
    uint32_t* s_flush = s_base + NB;
    uint32_t* s_nf = s_flush + 2 * NB;
    const uint32_t g = blockIdx.x & 7u;
    Rec* st = stage + (size_t)g * NB * BLK;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) *s_nf = 0;
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < NB; d += 1024) {
            const uint32_t p = atomicAdd(&scur[(size_t)g * NB + d], r);
            s_base[d] = p % BLK;
            if (p / BLK != (p + r) / BLK) {          // this append fills the block: flush it to the bin's next BLK slots
                const uint32_t k = atomicAdd(s_nf, 1u);
                s_flush[2 * k] = d;
                s_flush[2 * k + 1] = bucket_base[d] + atomicAdd(&cursor[(size_t)g * NB + d], BLK);
            }
        }
        __syncthreads();
        const uint32_t total = NB * r;
        for (uint32_t j = threadIdx.x; j < total; j += 1024) {
            const uint32_t d = j / r, o = j - d * r;
            uint32_t q = s_base[d] + o;
            if (q >= BLK) q -= BLK;
            st[(size_t)d * BLK + q] = Rec{j, t, d};
        }
        __syncthreads();                             // (the appends of THIS workgroup are visible to it; others' are not waited for)
        const uint32_t nf = *s_nf, ftotal = nf * BLK;
        for (uint32_t j = threadIdx.x; j < ftotal; j += 1024) {
            const uint32_t k = j / BLK, o = j - k * BLK;
            const Rec v = st[(size_t)s_flush[2 * k] * BLK + o];
            Rec* dst = out + (size_t)s_flush[2 * k + 1] + o;
            if (NT_STORE) {
                __builtin_nontemporal_store(v.a, &dst->a); __builtin_nontemporal_store(v.b, &dst->b); __builtin_nontemporal_store(v.c, &dst->c);
            } else *dst = v;
        }
    }
}

__global__ void __launch_bounds__(1024)
k_stream(Rec* __restrict__ out, uint64_t records)
{
    for (uint64_t j = (uint64_t)blockIdx.x * 1024 + threadIdx.x; j < records; j += (uint64_t)gridDim.x * 1024)
        out[j] = Rec{(uint32_t)j, 1u, 2u};
}

// ./scatter_write --floor <records> <bins> <run>: ONE number for bench.py's roofline record -- the store pattern of the partition
// kernel alone (r-record runs of 12-byte records over `bins` first digits, per-(group, bin) cursors claimed with returning atomics)
// from a resident grid of one workgroup per CU, as k_msd_part_text runs (its LDS allows no second); best of five.
static int floor_mode(uint64_t records, uint32_t NB, uint32_t r)
{
    Rec* out; uint32_t* cursor; uint32_t* base; uint32_t* gbar;
    CK(hipMalloc(&out, (records + (uint64_t)NB * 4096) * sizeof(Rec)));
    CK(hipMalloc(&cursor, (size_t)8 * NB * 4));
    CK(hipMalloc(&base, (size_t)NB * 4));
    CK(hipMalloc(&gbar, 8 * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const uint32_t grid = (uint32_t)(cus / 8 * 8);
    const uint32_t tiles = (uint32_t)(records / ((uint64_t)NB * r));
    std::vector<uint32_t> hb(NB);
    for (uint32_t d = 0; d < NB; d++) hb[d] = (uint32_t)((uint64_t)d * tiles * r);
    CK(hipMemcpy(base, hb.data(), (size_t)NB * 4, hipMemcpyHostToDevice));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        std::vector<uint32_t> hc((size_t)8 * NB, 0u);
        const uint32_t per_group = (tiles + 7) / 8 * r;
        for (uint32_t g = 0; g < 8; g++) for (uint32_t d = 0; d < NB; d++) hc[(size_t)g * NB + d] = g * per_group;
        CK(hipMemcpy(cursor, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(gbar, 0, 8 * 64 * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_scatter_lockstep, dim3(grid), dim3(1024), (size_t)NB * 4, 0, out, base, cursor, NB, r, tiles, gbar, 0u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("{\"store_floor_ms\": %.4f, \"records\": %llu, \"bins\": %u, \"run\": %u, \"grid\": %u, \"bytes\": %llu}\n", best,
           (unsigned long long)tiles * NB * r, NB, r, grid, (unsigned long long)tiles * NB * r * 12ull);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 4 && !strcmp(argv[1], "--floor"))
        return floor_mode(strtoull(argv[2], nullptr, 10), (uint32_t)atoi(argv[3]), (uint32_t)atoi(argv[4]));
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
    // ---- the reverse walk: scattered READS of r-record runs, contiguous writes ----
    {
        Rec* in; CK(hipMalloc(&in, records * sizeof(Rec)));
        CK(hipMemset(in, 1, records * sizeof(Rec)));
        for (uint32_t r : {6u, 12u, 24u}) {
            const uint32_t tiles = (uint32_t)(records / ((uint64_t)NB * r));
            for (uint32_t ggrid : {512u, 1024u, 2700u}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_gather_runs, dim3(ggrid), dim3(1024), 0, 0, out, in, NB, r, tiles);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                }
                printf("gather-runs bins %5u  run %4u rec (%5u B)  grid %4u: %7.2f ms (read %.1f GB scattered + write %.1f GB contiguous)\n", NB, r, r * 12, ggrid, best,
                       (double)tiles * NB * r * 12 / 1e9, (double)tiles * NB * r * 12 / 1e9);
            }
        }
        CK(hipFree(in));
    }
    // ---- the direct pattern, the workgroups of an XCD in lockstep ----
    {
        uint32_t* gbar; CK(hipMalloc(&gbar, 8 * 64 * 4));
        for (uint32_t r : {6u, 12u, 24u}) {
            const uint32_t tiles = (uint32_t)(records / ((uint64_t)NB * r));
            std::vector<uint32_t> hb(NB);
            for (uint32_t d = 0; d < NB; d++) hb[d] = (uint32_t)((uint64_t)d * tiles * r);
            CK(hipMemcpy(base, hb.data(), (size_t)NB * 4, hipMemcpyHostToDevice));
            for (uint32_t lgrid : {512u, 256u}) for (uint32_t nsync : {0u, 1u, 4u, 16u}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    std::vector<uint32_t> hc((size_t)8 * NB, 0u);
                    const uint32_t per_group = (tiles + 7) / 8 * r;
                    for (uint32_t g = 0; g < 8; g++) for (uint32_t d = 0; d < NB; d++) hc[(size_t)g * NB + d] = g * per_group;
                    CK(hipMemcpy(cursor, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
                    CK(hipMemset(gbar, 0, 8 * 64 * 4));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_scatter_lockstep, dim3(lgrid), dim3(1024), (size_t)NB * 4, 0, out, base, cursor, NB, r, tiles, gbar, nsync);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                }
                printf("lockstep bins %5u  run %4u rec (%5u B)  grid %3u (%u workgroups per XCD), %2u barriers per tile: %7.2f ms\n", NB, r, r * 12, lgrid, lgrid / 8, nsync, best);
            }
        }
    }
    // ---- XCD-local write combining: r = 12 records per append (the stand-in's pattern), blocks of BLK records ----
    {
        const uint32_t r = 12;
        const uint32_t tiles = (uint32_t)(records / ((uint64_t)NB * r));
        std::vector<uint32_t> hb(NB);
        for (uint32_t d = 0; d < NB; d++) hb[d] = (uint32_t)((uint64_t)d * tiles * r);
        CK(hipMemcpy(base, hb.data(), (size_t)NB * 4, hipMemcpyHostToDevice));
        uint32_t* scur; CK(hipMalloc(&scur, (size_t)8 * NB * 4));
        for (uint32_t BLK : {48u, 64u, 96u, 128u, 192u}) {
            Rec* stage; CK(hipMalloc(&stage, (size_t)8 * NB * BLK * sizeof(Rec)));
            for (int nt = 0; nt < 2; nt++) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    std::vector<uint32_t> hc((size_t)8 * NB, 0u);
                    const uint32_t per_group = (tiles + 7) / 8 * r;
                    for (uint32_t g = 0; g < 8; g++) for (uint32_t d = 0; d < NB; d++) hc[(size_t)g * NB + d] = g * per_group;
                    CK(hipMemcpy(cursor, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
                    CK(hipMemset(scur, 0, (size_t)8 * NB * 4));
                    CK(hipEventRecord(e0));
                    const size_t lds = (size_t)NB * 12 + 16;
                    if (nt) hipLaunchKernelGGL(k_combine<true>, dim3(grid), dim3(1024), lds, 0, out, stage, base, cursor, scur, NB, r, BLK, tiles);
                    else hipLaunchKernelGGL(k_combine<false>, dim3(grid), dim3(1024), lds, 0, out, stage, base, cursor, scur, NB, r, BLK, tiles);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                }
                printf("combine bins %5u  append %u rec, block %4u rec (%5u B, staging %.1f MB per XCD)%s: %7.2f ms\n", NB, r, BLK, BLK * 12,
                       NB * BLK * 12.0 / 1e6, nt ? " nt stores" : "          ", best);
            }
            CK(hipFree(stage));
        }
    }
    return 0;
}
