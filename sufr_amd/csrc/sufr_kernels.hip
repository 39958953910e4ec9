// sufr_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for suffix-array + LCP construction.
//
// Replaces the CPU hot loops of the reference builder (libsufr/src/sufr_builder.rs):
//   * text normalisation            (143-160)  -> k_text_pass_dna / k_normalize_bytehist (+ run-end tables, bit-packed
//                                                 code stream, suffix-start bitmap, first-digit histogram)
//   * eligibility + upper_bound/is_less/find_lcp bucketing (346-394, 442-462)
//                                              -> MSD partition levels on packed k-character prefix keys
//                                                 (sufr_part.inc: k_msd_part_text; sufr_msd.inc: k_msd_scatter; alphabets
//                                                 without a packed stream: k_hist_text + k_scatter_text here)
//   * merge_sort / merge            (601-767)  -> k_leaf_sort (sufr_msd.inc: in-LDS sort of a bucket, SA, LCP from
//                                                 the key xor) + the tie / re-keying levels here (k_gather_keys, the
//                                                 group sorts of sufr_msd.inc, k_plan_windows + k_finish: wave-level tie
//                                                 refinement, exact LCP) + prefix doubling (sufr_dbl.inc) + buckets of one
//                                                 repeated symbol placed by counting (sufr_runs.inc; find_lcp's byte walk
//                                                 through a run, 319-329, never happens)
//   * boundary LCP of write()       (886-906)  -> LCP at every group / window / shard boundary (key xor,
//                                                 k_fix_window_lcp, k_lcp_pair)
//
// Integer / byte work, HBM-bound: no MFMA.  64-wide wavefronts are hard-coded.
//
// Ordering contract (what the reference produces): suffixes compare as raw byte
// strings of the normalised text; a suffix that is a proper prefix of another sorts first;
// LCP[0] = 0, LCP[i] = exact common prefix of SA[i-1], SA[i].

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "sufr_device.h"
#include "sufr_runkey.h"

namespace sufr {
// phase stamps (probes build): cycles per phase of the last wave of every workgroup, summed; printed with SUFR_HIP_DEBUG=1
#ifdef SUFR_HIP_PROBES
__device__ unsigned long long g_phase[3][16];          // cycles per phase, summed over workgroups: part, scatter, k_finish<DEEP>
#define SUFR_STAMP(acc, i) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); acc[i] += t__ - stamp__; stamp__ = t__; }
#define SUFR_STAMP_DECL(n) unsigned long long ph__[n] = {}; unsigned long long stamp__ = __builtin_amdgcn_s_memtime();
#define SUFR_STAMP_FLUSH(k, n) if ((threadIdx.x & 63u) == 0 && (threadIdx.x >> 6) == gridDim.y * 0 + (blockDim.x >> 6) - 1) { for (int i__ = 0; i__ < n; i__++) atomicAdd(&g_phase[k][i__], ph__[i__]); }
#else
#define SUFR_STAMP(acc, i)
#define SUFR_STAMP_DECL(n)
#define SUFR_STAMP_FLUSH(k, n)
#endif
}  // namespace sufr

namespace sufr {

static constexpr int WAVE = 64;
static_assert(RUN_TILE == (uint32_t)TILE, "run-end table granularity");

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t rec_key(const Rec& r) { return ((uint64_t)r.khi << 32) | r.klo; }
__device__ __forceinline__ Rec make_rec(uint64_t key, uint32_t idx)
{
    Rec r; r.klo = (uint32_t)key; r.khi = (uint32_t)(key >> 32); r.idx = idx;
    return r;
}

// Pointer to LDS that stays in the LDS address space: accesses compile to ds_read/ds_write, which one wave
// issues and completes in order.  (A plain `volatile T*` to a __shared__ object becomes a generic pointer
// and hipcc then emits FLAT instructions, whose LDS accesses are NOT ordered -- a cross-lane exchange
// through them races.)
#define SUFR_LDS_VOLATILE(T, p) ((__attribute__((address_space(3))) volatile T*)(p))

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, WAVE);
    hi = __shfl(hi, src, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl64_xor(uint64_t v, int m)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, WAVE);
    hi = __shfl_xor(hi, m, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl64_up1(uint64_t v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_up(lo, 1, WAVE);
    hi = __shfl_up(hi, 1, WAVE);
    return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------------------
// k_normalize_bytehist: reference text map (sufr_builder.rs:144-160) fused with an exact
// 256-bin byte histogram of the NORMALISED text.  The histogram drives the alphabet -> dense code
// table, the eligible-suffix count and the pass-count heuristic.
// LDS histogram is replicated 32x (bin-major) so that lanes l and l+32 share a copy and every
// lane of a half-wave hits its own bank.  The same pass records, per 4096-byte tile, the first position
// where a run of equal bytes ends (first_end[], the RunTable of sufr_runkey.h).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_normalize_bytehist(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n,
                     int normalize, int ignore_softmask, unsigned long long* __restrict__ ghist,
                     uint32_t* __restrict__ first_end, uint64_t* __restrict__ run_ends,
                     uint64_t* __restrict__ tile_any)
{
    __shared__ uint32_t h[256 * 32];
    __shared__ uint8_t s_first[256 + 4];
    __shared__ uint32_t s_min;
    __shared__ uint32_t s_any[2];
    for (int i = threadIdx.x; i < 256 * 32; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t copy = threadIdx.x & 31u;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    auto norm = [&](uint32_t b) -> uint32_t {
        if (normalize && b >= 97u && b <= 122u) b = ignore_softmask ? 78u : (b & 0x5Fu);
        return b;
    };
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t p0 = tile * TILE + (uint64_t)threadIdx.x * 16;
        uint32_t by[17];
        if (threadIdx.x == 0) { s_min = 0xffffffffu; s_any[0] = 0; s_any[1] = 0; }
        if (p0 + 16 <= n) {
            uint4 w = *reinterpret_cast<const uint4*>(in + p0);
            uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t y = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t b = norm((ws[k] >> (8 * j)) & 0xffu);
                    by[4 * k + j] = b;
                    y |= b << (8 * j);
                    atomicAdd(&h[b * 32 + copy], 1u);
                }
                ws[k] = y;
            }
            *reinterpret_cast<uint4*>(out + p0) = make_uint4(ws[0], ws[1], ws[2], ws[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                by[e] = 0;
                if (p0 + e < n) {
                    uint32_t b = norm(in[p0 + e]);
                    by[e] = b;
                    out[p0 + e] = (uint8_t)b;
                    atomicAdd(&h[b * 32 + copy], 1u);
                }
            }
        }
        s_first[threadIdx.x] = (uint8_t)by[0];
        __syncthreads();
        // byte after this thread's 16: the next thread's first byte, or the next tile's first byte
        if (threadIdx.x < 255) by[16] = s_first[threadIdx.x + 1];
        else by[16] = (p0 + 16 < n) ? norm(in[p0 + 16]) : 0u;
        // first position of the tile where a run of equal bytes ends (RunTable::first_end)
        uint32_t best = 0xffffffffu;
        uint32_t ends16 = 0;
#pragma unroll
        for (int e = 15; e >= 0; e--) {
            uint64_t p = p0 + e;
            if (p < n && (p == n - 1 || by[e] != by[e + 1])) { best = (uint32_t)p; ends16 |= 1u << e; }
        }
        if (best != 0xffffffffu) atomicMin(&s_min, best);
        // run-end bitmap: four threads share a 64-bit word (bit i of word w <-> position 64 w + i)
        {
            uint64_t v = (uint64_t)ends16 << (16 * (threadIdx.x & 3u));
            v |= shfl64_xor(v, 1);
            v |= shfl64_xor(v, 2);
            if ((threadIdx.x & 3u) == 0) {
                const uint32_t j = threadIdx.x >> 2;
                run_ends[tile * 64 + j] = v;
                if (v) atomicOr(&s_any[j >> 5], 1u << (j & 31u));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            first_end[tile] = s_min;
            tile_any[tile] = (uint64_t)s_any[0] | ((uint64_t)s_any[1] << 32);
        }
        __syncthreads();
    }
    {
        uint32_t sum = 0;
        for (int c = 0; c < 32; c++) sum += h[threadIdx.x * 32 + ((c + threadIdx.x) & 31)];
        if (sum) atomicAdd(&ghist[threadIdx.x], (unsigned long long)sum);
    }
}

// Several buffers cleared by ONE launch (round 5): every hipMemsetAsync is a launch of its own -- ~8 us of the stream with its gap --
// and a build clears five small ranges before its first kernel and three before every leaf stage: 10 % of a 4.6 Mb build.
struct ZeroList { unsigned long long p[8]; unsigned long long bytes[8]; uint32_t n; };
struct ZeroSizes { unsigned long long bytes[8]; };
// (the pointers are kernel arguments of their own: pointers inside a by-value struct are generic, and a store through one is a
// flat_store -- tests/test_host_logic.py keeps flat instructions out of the kernels)
__device__ __forceinline__ void zero_range(uint8_t* __restrict__ p, size_t nb, size_t tid, size_t nth)
{
    uint8_t* e = p + nb;
    uint8_t* a = p + ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u);      // 16-byte body
    if (a > e) a = e;
    const size_t head = (size_t)(a - p), nu = (size_t)(e - a) / 16, tail = (size_t)(e - a) - nu * 16;
    for (size_t i = tid; i < head; i += nth) p[i] = 0;
    uint4* b = reinterpret_cast<uint4*>(a);
    for (size_t i = tid; i < nu; i += nth) b[i] = make_uint4(0u, 0u, 0u, 0u);
    uint8_t* ae = a + nu * 16;
    for (size_t i = tid; i < tail; i += nth) ae[i] = 0;
}
__global__ void __launch_bounds__(256)
k_zero_ranges(uint8_t* __restrict__ p0, uint8_t* __restrict__ p1, uint8_t* __restrict__ p2, uint8_t* __restrict__ p3,
              uint8_t* __restrict__ p4, uint8_t* __restrict__ p5, uint8_t* __restrict__ p6, uint8_t* __restrict__ p7, ZeroSizes z)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    if (z.bytes[0]) zero_range(p0, z.bytes[0], tid, nth);
    if (z.bytes[1]) zero_range(p1, z.bytes[1], tid, nth);
    if (z.bytes[2]) zero_range(p2, z.bytes[2], tid, nth);
    if (z.bytes[3]) zero_range(p3, z.bytes[3], tid, nth);
    if (z.bytes[4]) zero_range(p4, z.bytes[4], tid, nth);
    if (z.bytes[5]) zero_range(p5, z.bytes[5], tid, nth);
    if (z.bytes[6]) zero_range(p6, z.bytes[6], tid, nth);
    if (z.bytes[7]) zero_range(p7, z.bytes[7], tid, nth);
}

// Small results for the host, published by the device itself: the queued (source, size) pairs are copied into MAPPED pinned
// memory and a sequence number is stored behind them; the host spins on that word instead of paying for a copy command and a
// stream synchronisation (7 against 16 us per round trip, profiles/micro/readback.hip; Pipeline::sync_reads).
struct PubList { const uint32_t* src[8]; uint32_t words[8]; uint32_t off[8]; uint32_t n; };
__global__ void __launch_bounds__(256)
k_publish(PubList pl, uint32_t* __restrict__ host, unsigned long long* __restrict__ flag, unsigned long long seq)
{
    for (uint32_t r = 0; r < pl.n; r++)
        for (uint32_t i = threadIdx.x; i < pl.words[r]; i += 256)
            __builtin_nontemporal_store(__builtin_nontemporal_load(pl.src[r] + i), host + pl.off[r] + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// ---------------------------------------------------------------------------------------------
// k_text_pass_dna: THE one pass over the raw text of a DNA build (replaces the text map of SufrBuilder::new,
// sufr_builder.rs:144-160, and everything the pivot selection 771-809 needed to know about the text).
// Specialised for the alphabet a DNA build almost always has after normalisation, {'$', '%', 'A', 'C', 'G', 'N',
// 'T'}: with the code table fixed in advance (codes 1..7 in byte order, 3 bits; 0 = past the end) one read of the
// text yields
//   * the normalised text (what the .sufr file stores),
//   * the bit-packed code stream (3/8 byte per base) the partition kernel and the run keys read,
//   * the run-end tables (RunTable of sufr_runkey.h: bitmap, first run end and summary word per 4 KB tile),
//   * the eight symbol counts (counts[8] = bytes outside the set: the host then falls back to
//     k_normalize_bytehist and the general code table),
//   * the histogram of the first DIGIT (5 characters, 15 bits, raw values) of every suffix start, one row per
//     group of workgroups (group = blockIdx % ngroups, the same chunk -> group map as k_msd_part_text: the
//     write cursors of the partition kernel come from these rows), and
//   * the set of 5-mers that occur at ANY position (presence bits; suffix starts are in the histogram): the
//     dense digit numbering of all MSD levels, and
//   * the suffix-start bitmap (one bit per position: eligibility, sufr_builder.rs:446-449), the work list of
//     k_msd_part_text.
// Round 2 read the text three times for this (normalise + pack, presence, first-digit histogram).
//
// No workgroup barrier inside the loop: a wave owns a 4 KB tile (four sub-blocks of 64 lanes x 16 bytes, all
// four loads in flight, the next tile's fetched before this one is worked on), takes the four characters after a
// lane's 16 from the next lane (whole-wave DPP shift; the last lane from the next sub-block, the last
// sub-block's from memory) and stages its 384 packed bytes per sub-block through a wave-private piece of LDS.
// The 2^15 raw histogram counters are 128 KB of LDS: one 1024-thread workgroup per CU, 16 autonomous waves.
// Equal consecutive 5-mers of a lane (homopolymer runs: N runs are half of a soft-masked genome) are counted
// in a register and added once: an LDS atomic that 64 lanes aim at one counter is serialised.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dna_fixed_code(uint32_t b)
{
    uint32_t c = 0;                  // 0: not in the set
    c = b == 0x24u ? 1u : c;         // '$'
    c = b == 0x25u ? 2u : c;         // '%'
    c = b == 0x41u ? 3u : c;         // 'A'
    c = b == 0x43u ? 4u : c;         // 'C'
    c = b == 0x47u ? 5u : c;         // 'G'
    c = b == 0x4eu ? 6u : c;         // 'N'
    c = b == 0x54u ? 7u : c;         // 'T'
    return c;
}

static constexpr int TP_NT = 1024;                 // threads of k_text_pass_dna
static constexpr int TP_RAW_BINS = 1 << 15;        // raw values of a 5-character digit of 3-bit codes
static constexpr size_t TP_LDS = (size_t)TP_RAW_BINS * 4 + (TP_RAW_BINS / 32) * 4 + 256 * 2 + (TP_NT / 64) * 192 * 2 + 9 * 8;

// lane i takes the value of lane i + 1 of its wave (lane 63 keeps its own)
__device__ __forceinline__ uint32_t wave_next_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false);
}

__global__ void __launch_bounds__(TP_NT)
k_text_pass_dna(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n, int normalize,
                int ignore_softmask, unsigned long long* __restrict__ counts,
                uint32_t* __restrict__ first_end, uint64_t* __restrict__ run_ends,
                uint64_t* __restrict__ tile_any, uint8_t* __restrict__ packed, uint32_t elig_codes,
                uint64_t chunk, uint32_t ngroups, uint32_t* __restrict__ rawtab, uint32_t* __restrict__ presbits,
                uint64_t* __restrict__ startbits, uint32_t* __restrict__ exc_pos, uint8_t* __restrict__ exc_byte, uint32_t exc_cap)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem);                                   // TP_RAW_BINS
    uint32_t* s_pres = s_hist + TP_RAW_BINS;                                                // TP_RAW_BINS / 32
    uint16_t* s_tab = reinterpret_cast<uint16_t*>(s_pres + TP_RAW_BINS / 32);               // 256: byte << 8 | code
    uint16_t* s_packall = s_tab + 256;                                                      // 192 per wave
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(s_packall + (TP_NT / 64) * 192);   // 9
    for (int i = threadIdx.x; i < TP_RAW_BINS; i += TP_NT) s_hist[i] = 0;
    if (threadIdx.x < TP_RAW_BINS / 32) s_pres[threadIdx.x] = 0;
    if (threadIdx.x < 256) {
        uint32_t b = threadIdx.x;
        if (normalize && b >= 97u && b <= 122u) b = ignore_softmask ? 78u : (b & 0x5Fu);
        s_tab[threadIdx.x] = (uint16_t)((b << 8) | dna_fixed_code(b));
    }
    if (threadIdx.x < 9) s_tot[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ln = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    constexpr uint32_t NW = TP_NT / 64;
    uint16_t* s_pack = s_packall + (size_t)wv * 192;
    uint64_t acc_even = 0, acc_odd = 0;      // 16-bit lanes: codes 0,2,4,6 / 1,3,5,7; code 0 = bytes outside the set
    // A workgroup takes the chunks blockIdx.x, blockIdx.x + gridDim.x, ... (round 5: one chunk per workgroup paid the 128 KB of
    // LDS counters -- cleared, then flushed with atomics -- once per chunk: 0.22 -> 0.17 ms at 100 Mb).  The grid is a multiple of
    // ngroups (or one workgroup per chunk), so all chunks of a workgroup belong to one group: chunk c -> group c % ngroups, the
    // map of k_msd_part_text.
    for (uint64_t cb = blockIdx.x; cb * chunk < n; cb += gridDim.x) {
    const uint64_t c0 = cb * chunk, c1 = min(c0 + chunk, n);      // chunk: a multiple of TILE
    const uint64_t t1 = (c1 + TILE - 1) / TILE;
    auto fetch = [&](uint64_t tile, uint4 (&w)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t p = tile * TILE + (uint64_t)k * 1024 + (uint64_t)ln * 16;
            w[k] = make_uint4(0, 0, 0, 0);
            if (p + 16 <= n) w[k] = *reinterpret_cast<const uint4*>(in + p);
            else if (p < n) {
                uint32_t q[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; e++) if (p + e < n) q[e >> 2] |= (uint32_t)in[p + e] << (8 * (e & 3));
                w[k] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        }
    };
    uint4 wn[4];
    uint64_t tile = c0 / TILE + wv;
    if (tile < t1) fetch(tile, wn);
    for (; tile < t1; tile += NW) {
        uint4 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = wn[k];
        if (tile + NW < t1) fetch(tile + NW, wn);
        const uint64_t base = tile * TILE;
        // text map + codes of a sub-block (sub-block k + 1 is mapped before k is finished: k's last lane needs its
        // first characters)
        uint32_t y[4][4];
        uint64_t V[4];                       // 16 codes, 3 bits each, first highest
        uint32_t L[4];                       // what the lane before needs: first four codes | first byte << 12
        auto map_sub = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const uint32_t ws[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
            const uint64_t p = base + (uint64_t)k * 1024 + (uint64_t)ln * 16;
            uint32_t vh = 0, vl = 0, na = 0, nb = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t yy = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t t = s_tab[(ws[q] >> (8 * j)) & 0xffu];
                    const uint32_t b = t >> 8, c = t & 0xffu;
                    yy |= b << (8 * j);
                    if (q < 2) { vh = (vh << 3) | c; na += 1u << (4 * c); }
                    else { vl = (vl << 3) | c; nb += 1u << (4 * c); }
                }
                y[k][q] = yy;
            }
            if (p + 16 <= n) {
                if (((na | nb) & 0xfu) != 0u) {
                    // a byte outside the table among these 16 (an IUPAC code, another delimiter: a few dozen positions of a real
                    // assembly): the text of THIS build carries 'N' there (code 6, same eligibility) and the position joins the
                    // exception list -- the suffixes whose comparisons reached it are re-placed by exc_reinsert after the build
                    // (everything of these 16 bytes is computed again, so that nothing of the first attempt stays live beside it: the
                    // kernel sits at 127 of its 128 registers, and the forms that patch the first attempt spill)
                    vh = 0; vl = 0; na = 0; nb = 0;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        uint32_t t = s_tab[(ws[e >> 2] >> (8 * (e & 3))) & 0xffu];
                        if ((t & 0xffu) == 0u) {
                            const unsigned long long at = atomicAdd(&counts[9], 1ull);
                            if (at < (unsigned long long)exc_cap) { exc_pos[at] = (uint32_t)(p + e); exc_byte[at] = (uint8_t)(t >> 8); }
                            t = (0x4eu << 8) | 6u;
                        }
                        const uint32_t c = t & 0xffu;
                        if ((e & 3) == 0) y[k][e >> 2] = 0;
                        y[k][e >> 2] |= (t >> 8) << (8 * (e & 3));
                        if (e < 8) { vh = (vh << 3) | c; na += 1u << (4 * c); } else { vl = (vl << 3) | c; nb += 1u << (4 * c); }
                    }
                }
                *reinterpret_cast<uint4*>(out + p) = make_uint4(y[k][0], y[k][1], y[k][2], y[k][3]);
            } else {
                // the text ends inside these 16 bytes (or before them): nothing past the end is a character
                uint32_t m[4] = {0, 0, 0, 0};
                vh = 0; vl = 0; na = 0; nb = 0;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    uint32_t c = 0;
                    if (p + e < n) {
                        uint32_t t = s_tab[(ws[e >> 2] >> (8 * (e & 3))) & 0xffu];
                        if ((t & 0xffu) == 0u) {                                   // (see above)
                            const unsigned long long at = atomicAdd(&counts[9], 1ull);
                            if (at < (unsigned long long)exc_cap) { exc_pos[at] = (uint32_t)(p + e); exc_byte[at] = (uint8_t)(t >> 8); }
                            t = (0x4eu << 8) | 6u;
                        }
                        c = t & 0xffu;
                        m[e >> 2] |= (t >> 8) << (8 * (e & 3));
                        out[p + e] = (uint8_t)(t >> 8);
                        if (e < 8) na += 1u << (4 * c); else nb += 1u << (4 * c);
                    }
                    if (e < 8) vh = (vh << 3) | c; else vl = (vl << 3) | c;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) y[k][q] = m[q];
            }
            {
                const uint32_t ea = na & 0x0f0f0f0fu, oa = (na >> 4) & 0x0f0f0f0fu;     // 8-bit lanes
                const uint32_t eb = nb & 0x0f0f0f0fu, ob = (nb >> 4) & 0x0f0f0f0fu;
                const uint32_t e8 = ea + eb, o8 = oa + ob;                               // <= 16 per lane
                acc_even += (uint64_t)(e8 & 0xffu) | ((uint64_t)((e8 >> 8) & 0xffu) << 16) |
                            ((uint64_t)((e8 >> 16) & 0xffu) << 32) | ((uint64_t)(e8 >> 24) << 48);
                acc_odd += (uint64_t)(o8 & 0xffu) | ((uint64_t)((o8 >> 8) & 0xffu) << 16) |
                           ((uint64_t)((o8 >> 16) & 0xffu) << 32) | ((uint64_t)(o8 >> 24) << 48);
            }
            V[k] = ((uint64_t)vh << 24) | (uint64_t)vl;
            L[k] = (vh >> 12) | ((y[k][0] & 0xffu) << 12);
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
        map_sub(K0{});
        // the four characters after the tile (its last lane's look-ahead)
        uint32_t Ltail = 0;
        if (ln == 63u) {
            const uint64_t p = base + TILE;
            uint32_t c4 = 0, b0 = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t t = 0;
                if (p + e < n) { t = s_tab[in[p + e]]; if ((t & 0xffu) == 0u) t = (0x4eu << 8) | 6u; }     // (listed by the tile's owner)
                c4 = (c4 << 3) | (t & 0xffu);
                if (e == 0) b0 = t >> 8;
            }
            Ltail = c4 | (b0 << 12);
        }
        uint32_t fe = RUN_NONE;              // first run end of the tile
        uint64_t anyw = 0;                   // bit j: word j of the tile's run-end bitmap is not zero
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k == 0) map_sub(K1{}); else if (k == 1) map_sub(K2{}); else if (k == 2) map_sub(K3{});
            __builtin_amdgcn_sched_barrier(0);
            uint32_t Lx = wave_next_u32(L[k]);
            {
                const uint32_t nextfirst = k < 3 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L[k < 3 ? k + 1 : 3]) : Ltail;
                if (ln == 63u) Lx = nextfirst;
            }
            const uint32_t n12 = Lx & 0xfffu, nbyte = Lx >> 12;
            // run ends: position e ends a run iff its byte differs from the next one (the byte after the text is 0)
            uint32_t ends16 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t nxt = q < 3 ? y[k][q < 3 ? q + 1 : 3] : nbyte;
                const uint32_t d = y[k][q] ^ ((y[k][q] >> 8) | (nxt << 24));
                const uint32_t z = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;     // high bit of every non-zero byte
                ends16 |= (((z >> 7) * 0x10204080u) >> 28) << (4 * q);
            }
            {
                // positions past the end of the text end nothing (their bytes are 0 and so is what follows)
                uint64_t v = (uint64_t)ends16 << (16 * (ln & 3u));
                v |= shfl64_xor(v, 1);
                v |= shfl64_xor(v, 2);
                const uint64_t bal = __ballot(v != 0ull);
                if ((ln & 3u) == 0u) run_ends[tile * 64 + (uint32_t)k * 16 + (ln >> 2)] = v;
                if (bal) {
                    uint64_t x = bal & 0x1111111111111111ull;                     // one bit per word
                    x = (x | (x >> 3)) & 0x0303030303030303ull;
                    x = (x | (x >> 6)) & 0x000f000f000f000full;
                    x = (x | (x >> 12)) & 0x000000ff000000ffull;
                    x = (x | (x >> 24)) & 0xffffull;
                    anyw |= x << (16 * k);
                    if (fe == RUN_NONE) {
                        const int q = __builtin_ctzll(bal);                       // a lane of the first word with an end
                        const uint64_t vq = shfl64(v, q);
                        fe = (uint32_t)(base + (uint64_t)k * 1024 + (uint64_t)(q >> 2) * 64 + (uint32_t)__builtin_ctzll(vq));
                    }
                }
            }
            // packed stream: 6 bytes per lane, 384 per sub-block, written as 24 x 16 bytes
            {
#pragma unroll
                for (int h = 0; h < 3; h++) {
                    const uint32_t hh = (uint32_t)(V[k] >> (16 * (2 - h))) & 0xffffu;
                    s_pack[ln * 3 + h] = (uint16_t)((hh >> 8) | (hh << 8));       // big-endian byte order
                }
                __builtin_amdgcn_wave_barrier();
                if (ln < 24u) {
                    const uint4 q = reinterpret_cast<const uint4*>(s_pack)[ln];
                    reinterpret_cast<uint4*>(packed + tile * (uint64_t)(TILE * 3 / 8) + (uint64_t)k * 384)[ln] = q;
                }
                __builtin_amdgcn_wave_barrier();
            }
            // first digit of every suffix start -> histogram; 5-mers that start no suffix -> presence bits
            {
                const uint64_t X = (V[k] << 12) | (uint64_t)n12;                  // 20 codes
                uint32_t pm = 0xffffffffu, pc = 0;                                // pending 5-mer, its suffix starts
                uint32_t em = 0;                                                  // bit e: position e starts a suffix
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t mer = (uint32_t)(X >> (45 - 3 * e)) & 0x7fffu;
                    const uint32_t el = (elig_codes >> (mer >> 12)) & 1u;
                    em |= el << e;
                    if (mer == pm) pc += el;
                    else {
                        if (pc) atomicAdd(&s_hist[pm], pc);
                        else if (pm != 0xffffffffu) atomicOr(&s_pres[pm >> 5], 1u << (pm & 31u));
                        pm = mer; pc = el;
                    }
                }
                if (pc) atomicAdd(&s_hist[pm], pc);
                else atomicOr(&s_pres[pm >> 5], 1u << (pm & 31u));
                // suffix-start bitmap (the partition kernel's work list): four lanes share a 64-bit word
                uint64_t sv = (uint64_t)em << (16 * (ln & 3u));
                sv |= shfl64_xor(sv, 1);
                sv |= shfl64_xor(sv, 2);
                if ((ln & 3u) == 0u) startbits[tile * 64 + (uint32_t)k * 16 + (ln >> 2)] = sv;
            }
        }
        if (ln == 0u) { first_end[tile] = fe; tile_any[tile] = anyw; }
    }
    // eight symbol counts (code 0 = bytes outside the set): wave reduction, then one atomic per wave -- per chunk: the 16-bit
    // lanes of the accumulators hold one chunk's bytes of a lane
    uint32_t cnt[8];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cnt[2 * c] = (uint32_t)(acc_even >> (16 * c)) & 0xffffu;
        cnt[2 * c + 1] = (uint32_t)(acc_odd >> (16 * c)) & 0xffffu;
    }
    acc_even = 0; acc_odd = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        uint32_t v = cnt[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
        if (ln == 0 && v) atomicAdd(&s_tot[c == 0 ? 8 : c], (unsigned long long)v);
    }
    }   // (chunks of this workgroup)
    __syncthreads();
    if (threadIdx.x < 9 && s_tot[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_tot[threadIdx.x]);
    uint32_t* row = rawtab + (size_t)(blockIdx.x % ngroups) * TP_RAW_BINS;
    for (int i = threadIdx.x; i < TP_RAW_BINS; i += TP_NT) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&row[i], c);
    }
    if (threadIdx.x < TP_RAW_BINS / 32) {
        const uint32_t b = s_pres[threadIdx.x];
        if (b) atomicOr(&presbits[threadIdx.x], b);
    }
}

// the listed bytes (k_text_pass_dna: bytes outside the fixed table, built as 'N') back into the normalised text (sufr_exc.inc)
__global__ void __launch_bounds__(256)
k_exc_restore(uint8_t* __restrict__ text, const uint32_t* __restrict__ pos, const uint8_t* __restrict__ byte, uint32_t E)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < E) text[pos[i]] = byte[i];
}

// rawtot[v] = suffix starts whose first digit is v (all groups); bit v of occurs[] set iff the 5-mer v occurs anywhere
// (4 KB: what a single-GPU build reads back -- the counts themselves only decide shard ranges)
__global__ void __launch_bounds__(256)
k_fold_raw(const uint32_t* __restrict__ rawtab, uint32_t ngroups, const uint32_t* __restrict__ presbits,
           uint32_t* __restrict__ rawtot, unsigned long long* __restrict__ occurs)
{
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;          // TP_RAW_BINS is a multiple of 256
    uint32_t t = 0;
    for (uint32_t g = 0; g < ngroups; g++) t += rawtab[(size_t)g * TP_RAW_BINS + v];
    rawtot[v] = t;
    const uint64_t m = __ballot(t || ((presbits[v >> 5] >> (v & 31u)) & 1u));
    if (lane_id() == 0) occurs[v >> 6] = m;
}

// grouptab[g][d] = suffix starts of group g whose first digit has dense value d: the raw values
// [vals[d] << fold, (vals[d] + 1) << fold) (fold > 0: a digit of fewer characters than the histogram's five)
__global__ void __launch_bounds__(256)
k_dense_grouptab(const uint32_t* __restrict__ rawtab, uint32_t ngroups, const uint32_t* __restrict__ vals,
                 uint32_t NB, int fold, uint32_t* __restrict__ grouptab)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ngroups * NB) return;
    const uint32_t g = i / NB, d = i % NB;
    const uint32_t* row = rawtab + (size_t)g * TP_RAW_BINS + ((size_t)vals[d] << fold);
    uint32_t t = 0;
    for (uint32_t j = 0; j < (1u << fold); j++) t += row[j];
    grouptab[i] = t;
}

// ---------------------------------------------------------------------------------------------
// next_tile[t] = smallest t' >= t whose tile holds a run end (RunTable).  Backward "nearest set flag" over
// the tiles in three small steps: inside blocks of 256 tiles (ballots), over the block heads (one
// workgroup, in LDS), then the tiles that found nothing inside their block take their block's successor.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t nearest_at_or_after(bool has, uint32_t my_index, uint32_t* s_first /*[4]*/)
{
    // 256 threads; returns the index of the nearest thread >= this one (in thread order) with `has`, else RUN_NONE
    const uint64_t m = __ballot(has);
    const uint32_t ln = lane_id();
    const uint64_t at = m >> ln;
    uint32_t r = at ? my_index + (uint32_t)__builtin_ctzll(at) : RUN_NONE;
    if (ln == 0) s_first[threadIdx.x >> 6] = r;          // first in this wave (lane 0 sees the whole mask)
    __syncthreads();
    for (int w = (threadIdx.x >> 6) + 1; w < 4 && r == RUN_NONE; w++) r = s_first[w];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256)
k_next_tile_blocks(const uint32_t* __restrict__ first_end, uint32_t ntiles, uint32_t* __restrict__ next_tile,
                   uint32_t* __restrict__ block_first)
{
    __shared__ uint32_t s_first[4];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const bool has = t < ntiles && first_end[t] != RUN_NONE;
    const uint32_t r = nearest_at_or_after(has, t, s_first);
    if (t < ntiles) next_tile[t] = r;
    if (threadIdx.x == 0) block_first[blockIdx.x] = r;
}

// the three steps in one launch for short texts (<= 16 blocks of 256 tiles = 16 Mb): one workgroup walks the blocks from the
// end, carrying the nearest run-end tile found so far (two launches fewer on the critical path of a small build)
__global__ void __launch_bounds__(256)
k_next_tile_small(const uint32_t* __restrict__ first_end, uint32_t ntiles, uint32_t* __restrict__ next_tile)
{
    __shared__ uint32_t s_first[4];
    const uint32_t nblk = (ntiles + 255u) / 256u;
    uint32_t carry = RUN_NONE;                             // first tile with a run end in the blocks after b (uniform)
    for (int b = (int)nblk - 1; b >= 0; b--) {
        const uint32_t t = (uint32_t)b * 256u + threadIdx.x;
        const bool has = t < ntiles && first_end[t] != RUN_NONE;
        uint32_t r = nearest_at_or_after(has, t, s_first);
        const uint32_t block_first = s_first[0] != RUN_NONE ? s_first[0] : (s_first[1] != RUN_NONE ? s_first[1] : (s_first[2] != RUN_NONE ? s_first[2] : s_first[3]));
        if (r == RUN_NONE) r = carry;
        if (t < ntiles) next_tile[t] = r;
        if (block_first != RUN_NONE) carry = block_first;
        __syncthreads();                                   // (s_first is rewritten by the next block)
    }
}

__global__ void __launch_bounds__(256)
k_next_tile_heads(uint32_t* __restrict__ block_first, uint32_t nblocks)
{
    // in place: block_first[b] := first tile with a run end in the blocks AFTER b.  At most 4096 blocks
    // (2^32 text bytes / 4096 / 256): staged in LDS, one thread walks them from the end.
    __shared__ uint32_t s_b[4096];
    for (uint32_t i = threadIdx.x; i < nblocks; i += 256) s_b[i] = block_first[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t nxt = RUN_NONE;
        for (int64_t b = (int64_t)nblocks - 1; b >= 0; b--) {
            const uint32_t own = s_b[b];
            s_b[b] = nxt;
            if (own != RUN_NONE) nxt = own;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nblocks; i += 256) block_first[i] = s_b[i];
}

__global__ void __launch_bounds__(256)
k_next_tile_apply(uint32_t* __restrict__ next_tile, uint32_t ntiles, const uint32_t* __restrict__ block_after)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < ntiles && next_tile[t] == RUN_NONE) next_tile[t] = block_after[blockIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Key packing.  code(byte) in 1..sigma (dense rank of the byte among the bytes present in the text,
// so integer order of codes == raw byte order); 0 = "past the end of the text" (sorts lowest,
// sufr_builder.rs:372-379).  key = K codes of b bits, first character in the most significant bits.
// lut[byte] = code | 0x8000 if a suffix may START with that byte (eligibility, 446-449).
//
// Radix digits are dchars whole characters (dbits = b * dchars <= 12 bits).  Only a fraction of the
// 2^dbits nominal digit values occurs in a text (DNA: ~450 of 4096 4-mers), so digits are mapped
// through `remap` (monotone, built from the set of dchars-mers present) to a dense range [0, nbins):
// every per-bin LDS array, the bin scan and the ballot loop of the stable ranking shrink accordingly.
// ---------------------------------------------------------------------------------------------
struct TileKeys {
    uint64_t key[EPT];
    uint32_t elig;  // bit e set: position e of this thread is a suffix start inside [0,n)
};

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift, uint32_t raw_mask, const uint16_t* s_remap)
{
    uint32_t r = (uint32_t)(key >> shift) & raw_mask;
    return s_remap ? (uint32_t)s_remap[r] : r;     // s_remap: LDS copy, or the global table (L1-resident)
}

__device__ __forceinline__ void load_lut(const uint16_t* __restrict__ glut, uint16_t* s_lut)
{
    for (int i = threadIdx.x; i < 256; i += THREADS) s_lut[i] = glut[i];
}

__device__ __forceinline__ const uint16_t* load_remap(const uint16_t* __restrict__ gremap, uint16_t* s_remap,
                                                      uint32_t raw_bins)
{
    if (!gremap) return nullptr;
    for (uint32_t i = threadIdx.x; i < raw_bins; i += THREADS) s_remap[i] = gremap[i];
    return s_remap;
}

// ---- tile staging, B = bits per character known at compile time (fast path, B <= 7) --------------
// Each thread converts 16 raw bytes to code bytes (code | 0x80 if eligible, 0 past the end) and stores
// them with one 16-byte LDS write; keys are then built from 16-byte LDS reads held in registers, so the
// LDS sees no strided byte traffic.
__device__ __forceinline__ uint4 encode16(uint4 w, uint64_t pos0, uint64_t n, const uint16_t* s_lut)
{
    uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t y = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t l = s_lut[(ws[k] >> (8 * j)) & 0xffu];
            uint32_t c = (pos0 + k * 4 + j < n) ? ((l & 0x7fu) | ((l >> 8) & 0x80u)) : 0u;
            y |= c << (8 * j);
        }
        ws[k] = y;
    }
    return make_uint4(ws[0], ws[1], ws[2], ws[3]);
}

// `mine` = the 16 raw bytes of this thread's own positions, loaded one tile ahead by the caller
template <int B>
__device__ __forceinline__ void stage_code_tile(const uint8_t* __restrict__ text, uint64_t tile0, uint64_t n,
                                                const uint16_t* s_lut, uint8_t* s_code, uint4 mine)
{
    uint4* dst = reinterpret_cast<uint4*>(s_code);
    dst[threadIdx.x] = encode16(mine, tile0 + (uint64_t)threadIdx.x * 16, n, s_lut);
    if (threadIdx.x < HALO / 16) {   // halo: the first bytes of the next tile (text is padded)
        const int v = TILE / 16 + threadIdx.x;
        uint4 w = reinterpret_cast<const uint4*>(text + tile0)[v];
        dst[v] = encode16(w, tile0 + (uint64_t)v * 16, n, s_lut);
    }
}

template <int B>
__device__ __forceinline__ void build_tile_keys_fast(const uint8_t* s_code, TileKeys& tk)
{
    constexpr int K = 64 / B;
    constexpr int NV = (EPT + K + 15) / 16;
    constexpr int low = 64 - K * B;
    uint32_t w[NV * 4];
    const uint4* src = reinterpret_cast<const uint4*>(s_code + threadIdx.x * EPT);
#pragma unroll
    for (int v = 0; v < NV; v++) {
        uint4 q = src[v];
        w[4 * v] = q.x; w[4 * v + 1] = q.y; w[4 * v + 2] = q.z; w[4 * v + 3] = q.w;
    }
    uint64_t key = 0;
#pragma unroll
    for (int j = 0; j < K; j++) key = (key << B) | (uint64_t)((w[j >> 2] >> (8 * (j & 3))) & 0x7fu);
    key <<= low;
    tk.elig = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        tk.key[e] = key;
        if ((w[e >> 2] >> (8 * (e & 3))) & 0x80u) tk.elig |= 1u << e;
        uint64_t c = (w[(e + K) >> 2] >> (8 * ((e + K) & 3))) & 0x7fu;
        key = (key << B) | (c << low);
    }
}

// Keys straight from the bit-packed code stream (B <= 4): the E + K codes a thread needs are at most 128
// consecutive bits starting on a byte boundary (E * B is a multiple of 8), i.e. two unaligned 8-byte loads
// and one 128-bit funnel shift per key -- no LDS staging, no per-character lookups, 3/8 of a byte of
// global traffic per base.  Codes past the end of the text are 0 in the stream and never eligible.
template <int B, int E>
__device__ __forceinline__ void build_keys_packed_t(const uint8_t* __restrict__ packed, uint64_t pos0,
                                                    uint32_t elig_codes, uint64_t (&key)[E], uint32_t& elig)
{
    static_assert(B >= 2 && B <= 4 && (E * B) % 8 == 0 && (E + 64 / B) * B <= 128, "packed window");
    constexpr int K = 64 / B;
    constexpr uint64_t keep = ~0ull << (64 - K * B);
    const uint8_t* pp = packed + ((pos0 * B) >> 3);
    const uint64_t hi = __builtin_bswap64(load_u64_unaligned(pp));
    const uint64_t lo = __builtin_bswap64(load_u64_unaligned(pp + 8));
    elig = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int s = e * B;
        const uint64_t v = s ? ((hi << s) | (lo >> (64 - s))) : hi;
        key[e] = v & keep;
        elig |= ((elig_codes >> (uint32_t)(v >> (64 - B))) & 1u) << e;
    }
}

// ---- generic path (any b): raw bytes staged in LDS, per-byte lookups ---------------------------------
__device__ __forceinline__ void stage_text_tile(const uint8_t* __restrict__ text, uint64_t tile0,
                                                uint8_t* s_text)
{
    const uint4* src = reinterpret_cast<const uint4*>(text + tile0);
    uint4* dst = reinterpret_cast<uint4*>(s_text);
    for (int v = threadIdx.x; v < (TILE + HALO) / 16; v += THREADS) dst[v] = src[v];
}

__device__ __forceinline__ void build_tile_keys(const uint8_t* s_text, const uint16_t* s_lut,
                                                uint64_t tile0, uint64_t n, int b, int K, TileKeys& tk)
{
    const int p0 = threadIdx.x * EPT;
    const int low = 64 - K * b;
    uint64_t key = 0;
    for (int j = 0; j < K; j++) {
        uint64_t pos = tile0 + p0 + j;
        uint32_t c = pos < n ? (uint32_t)(s_lut[s_text[p0 + j]] & 0x3ffu) : 0u;
        key = (key << b) | c;
    }
    key <<= low;
    tk.elig = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        uint64_t pos = tile0 + p0 + e;
        tk.key[e] = key;
        if (pos < n && (s_lut[s_text[p0 + e]] & 0x8000u)) tk.elig |= 1u << e;
        uint64_t nx = pos + K;
        uint32_t c = nx < n ? (uint32_t)(s_lut[s_text[p0 + e + K]] & 0x3ffu) : 0u;
        key = (key << b) | ((uint64_t)c << low);
    }
}

template <int B>
__device__ __forceinline__ void tile_keys(const uint8_t* __restrict__ text, uint64_t tile0, uint64_t n,
                                          const uint16_t* s_lut, uint8_t* s_tile, const KeyParams& kp,
                                          TileKeys& tk, uint4 mine)
{
    if constexpr (B > 0) {
        stage_code_tile<B>(text, tile0, n, s_lut, s_tile, mine);
        __syncthreads();
        build_tile_keys_fast<B>(s_tile, tk);
    } else {
        stage_text_tile(text, tile0, s_tile);
        __syncthreads();
        build_tile_keys(s_tile, s_lut, tile0, n, kp.b, kp.K, tk);
    }
}

// ---------------------------------------------------------------------------------------------
// k_pack_codes: the text as a big-endian stream of B-bit codes (B <= 4: 16 codes fit one 64-bit word),
// zero past the end.  Run keys then fetch "the next ~20 characters" with one unaligned 9-byte read
// (make_run_key) instead of a per-character loop; 3/8 byte per base for DNA.
// ---------------------------------------------------------------------------------------------
template <int B>
__global__ void __launch_bounds__(256)
k_pack_codes(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
             uint8_t* __restrict__ packed)
{
    __shared__ uint16_t s_lut[256];
    __shared__ __align__(16) uint16_t s_pack[256 * B];
    s_lut[threadIdx.x] = glut[threadIdx.x];
    __syncthreads();
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * 16;
    uint4 w = *reinterpret_cast<const uint4*>(text + p0);     // text is padded
    uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    uint64_t V = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t byte = (ws[j >> 2] >> (8 * (j & 3))) & 0xffu;
        uint64_t c = (p0 + j < n) ? (uint64_t)(s_lut[byte] & 0x7fu) : 0ull;
        V = (V << B) | c;
    }
#pragma unroll
    for (int k = 0; k < B; k++) {
        uint32_t h = (uint32_t)(V >> (16 * (B - 1 - k))) & 0xffffu;
        s_pack[threadIdx.x * B + k] = (uint16_t)((h >> 8) | (h << 8));   // big-endian byte order
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(packed + (uint64_t)blockIdx.x * (TILE * B / 8));
    const uint4* src = reinterpret_cast<const uint4*>(s_pack);
    for (int v = threadIdx.x; v < TILE * B / 8 / 16; v += 256) dst[v] = src[v];
}

// ---------------------------------------------------------------------------------------------
// k_digit_presence: which dchars-mers (top-digit values of the key of EVERY text position) occur.
// flags[v] = 1 for every value seen; the host turns the flags into the dense `remap`.
// ---------------------------------------------------------------------------------------------
template <int B>
__global__ void __launch_bounds__(THREADS)
k_digit_presence(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
                 KeyParams kp, uint64_t chunk, uint32_t* __restrict__ flags, uint8_t* __restrict__ packed)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* s_flag = smem;                                              // raw_bins
    uint8_t* s_tile = smem + kp.raw_bins;                                // TILE + HALO
    uint16_t* s_lut = reinterpret_cast<uint16_t*>(s_tile + TILE + HALO);  // 256
    for (uint32_t i = threadIdx.x; i < kp.raw_bins; i += THREADS) s_flag[i] = 0;
    load_lut(glut, s_lut);
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk;
    const uint64_t c1 = min(c0 + chunk, n);
    uint4 nxt = reinterpret_cast<const uint4*>(text + c0)[threadIdx.x];
    for (uint64_t tile0 = c0; tile0 < c1; tile0 += TILE) {
        __syncthreads();
        const uint4 mine = nxt;
        if (tile0 + TILE < c1) nxt = reinterpret_cast<const uint4*>(text + tile0 + TILE)[threadIdx.x];
        TileKeys tk;
        tile_keys<B>(text, tile0, n, s_lut, s_tile, kp, tk, mine);
#pragma unroll
        for (int e = 0; e < EPT; e++)
            if (tile0 + threadIdx.x * EPT + e < n) s_flag[(uint32_t)(tk.key[e] >> kp.top_shift)] = 1;
        if constexpr (B > 0 && B <= 4) {
            // the same code bytes, bit-packed big-endian (what k_pack_codes would produce)
            if (packed) {
                uint4 q = reinterpret_cast<const uint4*>(s_tile)[threadIdx.x];
                uint32_t ws[4] = {q.x, q.y, q.z, q.w};
                uint64_t V = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) V = (V << B) | (uint64_t)((ws[j >> 2] >> (8 * (j & 3))) & 0x7fu);
                __syncthreads();                        // everyone has read its codes: reuse the tile buffer
                uint16_t* s_pack = reinterpret_cast<uint16_t*>(s_tile);
#pragma unroll
                for (int k = 0; k < B; k++) {
                    uint32_t hh = (uint32_t)(V >> (16 * (B - 1 - k))) & 0xffffu;
                    s_pack[threadIdx.x * B + k] = (uint16_t)((hh >> 8) | (hh << 8));
                }
                __syncthreads();
                uint4* dst = reinterpret_cast<uint4*>(packed + (tile0 / TILE) * (uint64_t)(TILE * B / 8));
                const uint4* src = reinterpret_cast<const uint4*>(s_pack);
                for (int v = threadIdx.x; v < TILE * B / 8 / 16; v += THREADS) dst[v] = src[v];
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kp.raw_bins; i += THREADS)
        if (s_flag[i]) flags[i] = 1u;
}

// ---------------------------------------------------------------------------------------------
// k_hist_text: per-workgroup digit histogram of the first radix pass, streaming the text.
// Row w of `table` (layout [workgroup][bin]) counts the digit of every eligible position of workgroup
// w's text chunk whose TOP digit lies in [top_lo, top_hi) (the shard filter used when a genome is split
// over GPUs by prefix-bucket range; top_lo/top_hi are RAW digit values: the dense remap is monotone, so a
// dense range is a raw range and the filter needs no table lookup).
// ---------------------------------------------------------------------------------------------
template <int B>
__global__ void __launch_bounds__(THREADS)
k_hist_text(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
            const uint16_t* __restrict__ gremap, KeyParams kp, int shift, uint64_t chunk,
            uint32_t top_lo, uint32_t top_hi, uint32_t tile_stride, uint32_t* __restrict__ table)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem);                  // nbins
    uint8_t* s_tile = smem + (size_t)kp.nbins * 4;                        // TILE + HALO
    uint16_t* s_lut = reinterpret_cast<uint16_t*>(s_tile + TILE + HALO);  // 256
    uint16_t* s_rm = s_lut + 256;                                         // raw_bins

    const uint32_t raw_mask = kp.raw_bins - 1;
    for (uint32_t i = threadIdx.x; i < kp.nbins; i += THREADS) s_cnt[i] = 0;
    load_lut(glut, s_lut);
    const uint16_t* s_remap = load_remap(gremap, s_rm, kp.raw_bins);
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk;
    const uint64_t c1 = min(c0 + chunk, n);
    const bool use_text = !(B >= 2 && B <= 4 && kp.packed);
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (use_text) nxt = reinterpret_cast<const uint4*>(text + c0)[threadIdx.x];
    for (uint64_t tile0 = c0; tile0 < c1; tile0 += (uint64_t)TILE * tile_stride) {
        if (use_text) __syncthreads();
        const uint4 mine = nxt;
        if (use_text && tile0 + (uint64_t)TILE * tile_stride < c1)
            nxt = reinterpret_cast<const uint4*>(text + tile0 + (uint64_t)TILE * tile_stride)[threadIdx.x];
        TileKeys tk;
        bool from_packed = false;
        if constexpr (B >= 2 && B <= 4) {
            if (kp.packed) {
                build_keys_packed_t<B, EPT>(kp.packed, tile0 + (uint64_t)threadIdx.x * EPT, kp.elig_codes, tk.key,
                                            tk.elig);
                from_packed = true;
            }
        }
        if (!from_packed) tile_keys<B>(text, tile0, n, s_lut, s_tile, kp, tk, mine);
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            if (tk.elig & (1u << e)) {
                uint32_t top = (uint32_t)(tk.key[e] >> kp.top_shift) & raw_mask;
                if (top >= top_lo && top < top_hi)
                    atomicAdd(&s_cnt[digit_of(tk.key[e], shift, raw_mask, s_remap)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * kp.nbins;
    for (uint32_t i = threadIdx.x; i < kp.nbins; i += THREADS) row[i] = s_cnt[i];
}

// ---------------------------------------------------------------------------------------------
// Block-wide exclusive scan of s_cnt[0..nbins).  On return s_cnt holds the exclusive prefix and
// *s_total the total.  Also applies the per-tile bookkeeping of the scatter kernels:
//     gdelta[d] = gbase[d] - prefix[d];   gbase[d] += count[d].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_scan_bins(uint32_t* s_cnt, uint32_t* s_gbase, uint32_t* s_gdelta,
                                                uint32_t nbins, uint32_t* s_wsum, uint32_t* s_total)
{
    const uint32_t per = (nbins + THREADS - 1) / THREADS;  // bins per thread (consecutive)
    const uint32_t d0 = threadIdx.x * per;
    uint32_t local = 0;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t d = d0 + k;
        if (d < nbins) local += s_cnt[d];
    }
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(incl, o, WAVE);
        if ((int)lane_id() >= o) incl += t;
    }
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += s_wsum[w];
    if (threadIdx.x == THREADS - 1) *s_total = wbase + incl;
    uint32_t run = wbase + incl - local;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t d = d0 + k;
        if (d < nbins) {
            uint32_t c = s_cnt[d];
            s_cnt[d] = run;
            uint32_t g = s_gbase[d];
            s_gdelta[d] = g - run;
            s_gbase[d] = g + c;
            run += c;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Geometry-generic helpers for the scatter kernels: NT threads per workgroup, E records (or text
// positions) per thread, tile = NT * E.  (The histogram kernels keep 256 x 16; only the workgroup -> chunk
// mapping has to agree, and chunks are multiples of every tile size.)
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void block_scan_bins_t(uint32_t* s_cnt, uint32_t* s_gbase, uint32_t* s_gdelta,
                                                  uint32_t nbins, uint32_t* s_wsum, uint32_t* s_total)
{
    const uint32_t per = (nbins + NT - 1) / NT;  // bins per thread (consecutive)
    const uint32_t d0 = threadIdx.x * per;
    uint32_t local = 0;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t d = d0 + k;
        if (d < nbins) local += s_cnt[d];
    }
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(incl, o, WAVE);
        if ((int)lane_id() >= o) incl += t;
    }
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += s_wsum[w];
    if (threadIdx.x == NT - 1) *s_total = wbase + incl;
    uint32_t run = wbase + incl - local;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t d = d0 + k;
        if (d < nbins) {
            uint32_t c = s_cnt[d];
            s_cnt[d] = run;
            uint32_t g = s_gbase[d];
            s_gdelta[d] = g - run;
            s_gbase[d] = g + c;
            run += c;
        }
    }
    __syncthreads();
}

template <int B, int E>
__device__ __forceinline__ void build_keys_t(const uint8_t* s_code, uint64_t (&key)[E], uint32_t& elig)
{
    constexpr int K = 64 / B;
    constexpr int NW8 = (E + K + 7) / 8;      // 8-byte words covering this thread's E + K code bytes
    constexpr int low = 64 - K * B;
    uint32_t w[NW8 * 2];
    const uint2* src = reinterpret_cast<const uint2*>(s_code + threadIdx.x * E);   // E % 8 == 0
#pragma unroll
    for (int v = 0; v < NW8; v++) {
        uint2 q = src[v];
        w[2 * v] = q.x; w[2 * v + 1] = q.y;
    }
    uint64_t k = 0;
#pragma unroll
    for (int j = 0; j < K; j++) k = (k << B) | (uint64_t)((w[j >> 2] >> (8 * (j & 3))) & 0x7fu);
    k <<= low;
    elig = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        key[e] = k;
        if ((w[e >> 2] >> (8 * (e & 3))) & 0x80u) elig |= 1u << e;
        uint64_t c = (w[(e + K) >> 2] >> (8 * ((e + K) & 3))) & 0x7fu;
        k = (k << B) | (c << low);
    }
}

// generic-alphabet variant: raw bytes staged, per-byte lookups
template <int E>
__device__ __forceinline__ void build_keys_generic_t(const uint8_t* s_text, const uint16_t* s_lut, uint64_t tile0,
                                                     uint64_t n, int b, int K, uint64_t (&key)[E], uint32_t& elig)
{
    const int p0 = threadIdx.x * E;
    const int low = 64 - K * b;
    uint64_t k = 0;
    for (int j = 0; j < K; j++) {
        uint64_t pos = tile0 + p0 + j;
        uint32_t c = pos < n ? (uint32_t)(s_lut[s_text[p0 + j]] & 0x3ffu) : 0u;
        k = (k << b) | c;
    }
    k <<= low;
    elig = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        uint64_t pos = tile0 + p0 + e;
        key[e] = k;
        if (pos < n && (s_lut[s_text[p0 + e]] & 0x8000u)) elig |= 1u << e;
        uint64_t nx = pos + K;
        uint32_t c = nx < n ? (uint32_t)(s_lut[s_text[p0 + e + K]] & 0x3ffu) : 0u;
        k = (k << b) | ((uint64_t)c << low);
    }
}

// ---------------------------------------------------------------------------------------------
// k_scatter_text: THE radix-partition kernel (first pass; suffix indices are implicit).
// Streams the text once (coalesced 16 B / lane), builds the packed key of every eligible suffix,
// ranks the tile's elements per digit with LDS counters, stages (key, idx) in LDS in digit order and
// writes every digit's run to its slot of the global bucket, so global stores are contiguous runs.
// Algorithmic bytes: n (text) + 4 s (indices) [+ 8 s for the keys carried to later passes].
// The first LSD pass has no earlier order to preserve, so ranking by LDS atomics is sufficient.
// ---------------------------------------------------------------------------------------------
template <int B, int NT, int E, bool SHARDED>
__global__ void __launch_bounds__(NT)
k_scatter_text(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
               const uint16_t* __restrict__ gremap, KeyParams kp, int shift, uint64_t chunk,
               uint32_t top_lo, uint32_t top_hi,
               const uint32_t* __restrict__ table, const uint32_t* __restrict__ binbase,
               Rec* __restrict__ out)
{
    constexpr int TILEB = NT * E;
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t NB = kp.nbins;
    const uint32_t NBa = (NB + 3u) & ~3u;                  // keep the arrays below 16-byte aligned
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem);   // NB
    uint32_t* s_gbase = s_cnt + NBa;                       // NB
    uint32_t* s_gdelta = s_gbase + NBa;                    // NB
    uint64_t* s_key = reinterpret_cast<uint64_t*>(s_gdelta + NBa);   // TILEB
    uint32_t* s_idx = reinterpret_cast<uint32_t*>(s_key + TILEB);    // TILEB
    uint8_t* s_tile = reinterpret_cast<uint8_t*>(s_idx + TILEB);     // TILEB + HALO
    uint16_t* s_lut = reinterpret_cast<uint16_t*>(s_tile + TILEB + HALO);  // 256
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(s_lut + 256);     // 32
    uint16_t* s_rm = reinterpret_cast<uint16_t*>(s_misc + 32);       // raw_bins

    const uint32_t raw_mask = kp.raw_bins - 1;
    const uint32_t* row = table + (size_t)blockIdx.x * NB;
    for (uint32_t i = threadIdx.x; i < NB; i += NT) s_gbase[i] = row[i] + binbase[i];
    for (int i = threadIdx.x; i < 256; i += NT) s_lut[i] = glut[i];
    // (tables stay in LDS: reading them through L1 instead measured slower, and a pointer that may be
    //  either LDS or global turns every lookup into a FLAT access)
    const uint16_t* s_remap = nullptr;
    if (gremap) {
        for (uint32_t i = threadIdx.x; i < kp.raw_bins; i += NT) s_rm[i] = gremap[i];
        s_remap = s_rm;
    }
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk;
    const uint64_t c1 = min(c0 + chunk, n);
    for (uint64_t tile0 = c0; tile0 < c1; tile0 += TILEB) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < NB; i += NT) s_cnt[i] = 0;
        uint64_t key[E];
#pragma unroll
        for (int e = 0; e < E; e++) key[e] = 0;
        uint32_t elig = 0;
        bool from_packed = false;
        if constexpr (B >= 2 && B <= 4) {
            if (kp.packed) {
                build_keys_packed_t<B, E>(kp.packed, tile0 + (uint64_t)threadIdx.x * E, kp.elig_codes, key, elig);
                from_packed = true;
            }
        }
        if (!from_packed) {
            // stage the tile (+ halo): raw bytes -> code bytes, one 16-byte LDS store per 16 positions
            const uint4* src = reinterpret_cast<const uint4*>(text + tile0);
            uint4* dst = reinterpret_cast<uint4*>(s_tile);
            for (int v = threadIdx.x; v < (TILEB + HALO) / 16; v += NT) {
                if constexpr (B > 0) dst[v] = encode16(src[v], tile0 + (uint64_t)v * 16, n, s_lut);
                else dst[v] = src[v];
            }
        }
        __syncthreads();
        if (from_packed) {
        } else if constexpr (B > 0) {
            // whole threads (and often whole waves) sit inside runs of ineligible bytes ('N' runs are half
            // of a soft-masked genome): peek at the eligibility bits before building any key
            const uint2* own = reinterpret_cast<const uint2*>(s_tile + threadIdx.x * E);
            uint32_t any = 0;
#pragma unroll
            for (int v = 0; v < E / 8; v++) { uint2 q = own[v]; any |= (q.x | q.y) & 0x80808080u; }
            if (any) build_keys_t<B, E>(s_tile, key, elig);
        } else {
            build_keys_generic_t<E>(s_tile, s_lut, tile0, n, kp.b, kp.K, key, elig);
        }
        // digit lookups for all E positions first (independent LDS reads in flight together), then the
        // ranking atomics under the eligibility mask: no load -> atomic dependency inside a branch
        uint32_t rank[E], dig[E];
        uint32_t keep = elig;
#pragma unroll
        for (int e = 0; e < E; e++) dig[e] = digit_of(key[e], shift, raw_mask, s_remap);
        if constexpr (SHARDED) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t top = (uint32_t)(key[e] >> kp.top_shift) & raw_mask;
                if (!(top >= top_lo && top < top_hi)) keep &= ~(1u << e);
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) rank[e] = (keep & (1u << e)) ? atomicAdd(&s_cnt[dig[e]], 1u) : 0u;
        // keys go to LDS linearly (own slots: conflict-free 16-byte stores); the suffix index is implicit
        // in the slot, so only a 4-byte (slot, digit) entry is scattered to the record's ranked position
        if (keep) {
            uint4* kd = reinterpret_cast<uint4*>(s_key + threadIdx.x * E);
#pragma unroll
            for (int v = 0; v < E / 2; v++)
                kd[v] = make_uint4((uint32_t)key[2 * v], (uint32_t)(key[2 * v] >> 32), (uint32_t)key[2 * v + 1],
                                   (uint32_t)(key[2 * v + 1] >> 32));
        }
        __syncthreads();
        block_scan_bins_t<NT>(s_cnt, s_gbase, s_gdelta, NB, s_misc, s_misc + 24);
        const uint32_t total = s_misc[24];
#pragma unroll
        for (int e = 0; e < E; e++) {
            if (keep & (1u << e)) {
                uint32_t pos = s_cnt[dig[e]] + rank[e];
                s_idx[pos] = ((uint32_t)(threadIdx.x * E + e) << 12) | dig[e];     // s_idx doubles as the permutation
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < total; j += NT) {
            const uint32_t pe = s_idx[j];
            const uint32_t slot = pe >> 12, d = pe & 0xfffu;
            const uint32_t o = j + s_gdelta[d];
            out[o] = make_rec(s_key[slot], (uint32_t)(tile0 + slot));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_hist_pairs: per-workgroup digit histogram for the later LSD passes.
// digit source: key bits (seg == nullptr) or the segment ordinal (deep levels).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
k_hist_pairs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ seg, uint32_t m,
             uint32_t nbins, uint32_t raw_bins, const uint16_t* __restrict__ gremap, int shift, uint32_t chunk,
             uint32_t* __restrict__ table)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem);                // nbins
    uint16_t* s_rm = reinterpret_cast<uint16_t*>(s_cnt + nbins);        // raw_bins
    const uint32_t mask = raw_bins - 1;
    for (uint32_t i = threadIdx.x; i < nbins; i += THREADS) s_cnt[i] = 0;
    const uint16_t* s_remap = load_remap(gremap, s_rm, raw_bins);
    __syncthreads();
    const uint32_t c0 = blockIdx.x * chunk;
    const uint32_t c1 = min(c0 + chunk, m);
    if (seg) {
        for (uint32_t j = c0 + threadIdx.x; j < c1; j += THREADS)
            atomicAdd(&s_cnt[(seg[j] >> shift) & mask], 1u);
    } else {
        // four 16-byte loads (two keys each) in flight per lane: 16 waves per CU then keep ~64 KB of reads
        // outstanding, which a streaming read needs to approach the HBM rate
        const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(keys + c0);   // c0 is a multiple of 8192
        const uint32_t npair = (c1 - c0) >> 1;
        uint32_t i = threadIdx.x;
        for (; i + 3u * THREADS < npair; i += 4u * THREADS) {
            ulonglong2 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = k2[i + (uint32_t)u * THREADS];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                atomicAdd(&s_cnt[digit_of(q[u].x, shift, mask, s_remap)], 1u);
                atomicAdd(&s_cnt[digit_of(q[u].y, shift, mask, s_remap)], 1u);
            }
        }
        for (; i < npair; i += THREADS) {
            const ulonglong2 q = k2[i];
            atomicAdd(&s_cnt[digit_of(q.x, shift, mask, s_remap)], 1u);
            atomicAdd(&s_cnt[digit_of(q.y, shift, mask, s_remap)], 1u);
        }
        if (((c1 - c0) & 1u) && threadIdx.x == 0)
            atomicAdd(&s_cnt[digit_of(keys[c1 - 1], shift, mask, s_remap)], 1u);
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * nbins;
    for (uint32_t i = threadIdx.x; i < nbins; i += THREADS) row[i] = s_cnt[i];
}

// ---------------------------------------------------------------------------------------------
// wave-level "match any": mask of the lanes whose digit equals this lane's digit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t match_digit(uint32_t d, int nbits, bool valid)
{
    uint64_t m = __ballot(valid);
    for (int k = 0; k < nbits; k++) {
        bool bit = (d >> k) & 1u;
        uint64_t bk = __ballot(bit);
        m &= bit ? bk : ~bk;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// k_scatter_pairs: stable LSD scatter of (key, idx [, seg]) records by one digit.
// Tile = THREADS*EPT records, striped per wave (record = strip0 + round*64 + lane), so that rank
// order == input order: rank = (#same digit in earlier rounds of this wave)   [per-wave LDS counter]
//                              + (#same digit in lower lanes of this round)    [ballot match]
//                              + (#same digit in earlier waves of the tile)    [prefix over waves]
// Records are staged in LDS in digit order and written out as contiguous runs per digit.
// ---------------------------------------------------------------------------------------------
// Lanes holding the same digit as this lane, as two 32-bit half masks.  Per bit: t = -bit (one v_bfe_i32),
// one ballot, and per half an xnor + and:  same_k = ~(ballot_k ^ t).
template <int NBITS>
__device__ __forceinline__ void match_digit_t(uint32_t d, int nbits, bool valid, uint32_t& mlo, uint32_t& mhi)
{
    const uint64_t vm = __ballot(valid);
    mlo = (uint32_t)vm; mhi = (uint32_t)(vm >> 32);
    auto step = [&](int k) {
        const int32_t t = ((int32_t)(d << (31 - k))) >> 31;      // 0 or -1
        const uint64_t bk = __ballot(t != 0);
        mlo &= ~((uint32_t)bk ^ (uint32_t)t);
        mhi &= ~((uint32_t)(bk >> 32) ^ (uint32_t)t);
    };
    if constexpr (NBITS > 0) {
#pragma unroll
        for (int k = 0; k < NBITS; k++) step(k);
    } else {
        for (int k = 0; k < nbits; k++) step(k);
    }
}

template <bool HAS_SEG, int NT, int E, int NBITS>
__global__ void __launch_bounds__(NT)
k_scatter_pairs(const uint64_t* __restrict__ in_key, const uint32_t* __restrict__ in_idx,
                const uint32_t* __restrict__ in_seg, uint32_t m, uint32_t nbins, int nbits,
                uint32_t raw_bins, const uint16_t* __restrict__ gremap,
                int shift, int digit_from_seg, uint32_t chunk,
                const uint32_t* __restrict__ table, const uint32_t* __restrict__ binbase,
                uint64_t* __restrict__ out_key, uint32_t* __restrict__ out_idx,
                uint32_t* __restrict__ out_seg)
{
    constexpr int TILEB = NT * E;
    constexpr int NW = NT / WAVE;
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t NB = nbins;
    const uint32_t NBa = (NB + 3u) & ~3u;
    uint32_t* s_gbase = reinterpret_cast<uint32_t*>(smem);  // NB
    uint32_t* s_gdelta = s_gbase + NBa;                     // NB
    uint32_t* s_tot = s_gdelta + NBa;                       // NB   tile count -> tile prefix
    uint32_t* s_misc = s_tot + NBa;                         // 32
    uint16_t* s_rm = reinterpret_cast<uint16_t*>(s_misc + 32);          // raw_bins (if remapped)
    uint8_t* s_union = reinterpret_cast<uint8_t*>(s_rm + (gremap ? ((raw_bins + 7u) & ~7u) : 0u));
    // union region: per-wave counters (NW*NB u16) during ranking, then the staging arrays
    uint16_t* s_wcnt = reinterpret_cast<uint16_t*>(s_union);
    uint64_t* s_key = reinterpret_cast<uint64_t*>(s_union);
    uint32_t* s_idx = reinterpret_cast<uint32_t*>(s_key + TILEB);
    uint32_t* s_seg = s_idx + TILEB;

    const uint32_t mask = raw_bins - 1;
    const uint32_t* row = table + (size_t)blockIdx.x * NB;
    for (uint32_t i = threadIdx.x; i < NB; i += NT) s_gbase[i] = row[i] + binbase[i];
    const uint16_t* s_remap = nullptr;
    if (gremap) {
        for (uint32_t i = threadIdx.x; i < raw_bins; i += NT) s_rm[i] = gremap[i];
        s_remap = s_rm;
    }
    const uint32_t c0 = blockIdx.x * chunk;
    const uint32_t c1 = min(c0 + chunk, m);
    const int wv = threadIdx.x >> 6;
    const uint32_t ln = lane_id();

    for (uint32_t tile0 = c0; tile0 < c1; tile0 += TILEB) {
        __syncthreads();  // previous tile's copy-out done before the union region is reused
        for (uint32_t i = threadIdx.x; i < (NW * NB + 1) / 2; i += NT)
            reinterpret_cast<uint32_t*>(s_wcnt)[i] = 0;
        __syncthreads();
        uint64_t key[E];
        uint32_t idx[E], sg[E], dig[E], rank[E];
        uint32_t valid = 0;
        const uint32_t strip0 = tile0 + wv * (E * WAVE);
#pragma unroll
        for (int e = 0; e < E; e++) {
            uint32_t j = strip0 + e * WAVE + ln;
            if (j < c1) {
                valid |= 1u << e;
                key[e] = in_key[j];
                idx[e] = in_idx[j];
                sg[e] = HAS_SEG ? in_seg[j] : 0u;
            } else {
                key[e] = 0; idx[e] = 0; sg[e] = 0;
            }
        }
        auto my = SUFR_LDS_VOLATILE(uint16_t, s_wcnt + (size_t)wv * NB);
        // digits of all E records first (independent LDS look-ups in flight together) ...
#pragma unroll
        for (int e = 0; e < E; e++)
            dig[e] = digit_from_seg ? ((sg[e] >> shift) & mask) : digit_of(key[e], shift, mask, s_remap);
        // ... then E straight-line ranking rounds (records strip0 + e*64 + lane)
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool v = valid & (1u << e);
            const uint32_t d = dig[e];
            uint32_t mlo, mhi;
            match_digit_t<NBITS>(d, nbits, v, mlo, mhi);
            const uint32_t before = my[d];                 // lanes without a record read a harmless slot
            // peers in lower lanes: v_mbcnt counts the mask bits below this lane directly
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            rank[e] = before + lower;
            if (v && lower == 0) my[d] = (uint16_t)(before + __popc(mlo) + __popc(mhi));
        }
        __syncthreads();
        // per digit: exclusive prefix over waves (in place) and tile count
        for (uint32_t d = threadIdx.x; d < NB; d += NT) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                uint32_t c = s_wcnt[(size_t)w * NB + d];
                s_wcnt[(size_t)w * NB + d] = (uint16_t)run;
                run += c;
            }
            s_tot[d] = run;
        }
        __syncthreads();
        block_scan_bins_t<NT>(s_tot, s_gbase, s_gdelta, NB, s_misc, s_misc + 24);
        const uint32_t total = s_misc[24];
        // final tile-local position of every record (registers), before the union region is reused
#pragma unroll
        for (int e = 0; e < E; e++)
            if (valid & (1u << e)) rank[e] += s_tot[dig[e]] + my[dig[e]];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; e++) {
            if (valid & (1u << e)) {
                s_key[rank[e]] = key[e];
                s_idx[rank[e]] = idx[e];
                if (HAS_SEG) s_seg[rank[e]] = sg[e];
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < total; j += NT) {
            uint64_t k = s_key[j];
            uint32_t sgv = HAS_SEG ? s_seg[j] : 0u;
            uint32_t d = digit_from_seg ? ((sgv >> shift) & mask) : digit_of(k, shift, mask, s_remap);
            uint32_t o = j + s_gdelta[d];
            out_key[o] = k;
            out_idx[o] = s_idx[j];
            if (HAS_SEG) out_seg[o] = sgv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Scan of the [workgroup][bin] histogram table.
//   k_scan_table_cols: per bin, exclusive prefix over workgroups (in place) + bin totals.
//   k_scan_bins:       exclusive prefix over the bin totals (single workgroup) + grand total.
// A record of workgroup w with digit d goes to  binbase[d] + table[w][d] + (rank inside w).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_scan_table_cols(uint32_t* __restrict__ table, uint32_t nwg, uint32_t nbins,
                  uint32_t* __restrict__ bintot)
{
    uint32_t d = blockIdx.x * 256 + threadIdx.x;
    if (d >= nbins) return;
    uint32_t run = 0;
    uint32_t w = 0;
    for (; w + 8 <= nwg; w += 8) {
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = table[(size_t)(w + k) * nbins + d];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            table[(size_t)(w + k) * nbins + d] = run;
            run += c[k];
        }
    }
    for (; w < nwg; w++) {
        uint32_t c = table[(size_t)w * nbins + d];
        table[(size_t)w * nbins + d] = run;
        run += c;
    }
    bintot[d] = run;
}

__global__ void __launch_bounds__(256)
k_scan_bins(const uint32_t* __restrict__ bintot, uint32_t nbins, uint32_t* __restrict__ binbase,
            unsigned long long* __restrict__ total_out)
{
    __shared__ uint32_t s_w[4];
    const uint32_t per = (nbins + 255) / 256;
    const uint32_t d0 = threadIdx.x * per;
    uint32_t local = 0;
    for (uint32_t k = 0; k < per; k++)
        if (d0 + k < nbins) local += bintot[d0 + k];
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(incl, o, WAVE);
        if ((int)lane_id() >= o) incl += t;
    }
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += s_w[w];
    uint32_t run = wbase + incl - local;
    for (uint32_t k = 0; k < per; k++) {
        if (d0 + k < nbins) {
            binbase[d0 + k] = run;
            run += bintot[d0 + k];
        }
    }
    if (threadIdx.x == 255 && total_out) *total_out = (unsigned long long)run;
}

// ---------------------------------------------------------------------------------------------
// k_finish: wave-level finisher.  Input: records sorted by (segment, top `sorted_bits` of the key).
// A *group* is a maximal run of records equal in (segment, those bits).  Wave t owns the window of
// records [128t, 128t+128) and the groups whose first record lies in it.
//   * groups that end inside the window ("small") are completely ordered here: round 0 ranks the
//     members by their whole key; every further round re-keys the still-tied suffixes with a run key
//     taken where their common prefix ends and ranks again, until every suffix is alone.  Ranking is a
//     counting sort through wave-private LDS (rank = members that compare lower), cheap for the tiny
//     groups that dominate; the LCP written at a split is depth + common characters of the two keys.
//   * groups that run past the window's end ("large") are appended to `large_heads`; the next level re-keys
//     them where their common prefix ends (k_group_extent / k_build_level / k_gather_keys).
// The LCP at a group's first record is its key-derived LCP with the record before it (another group):
// the boundary LCP the reference recomputes in write() (sufr_builder.rs:893-902).
// Level 0 keys are plain packed characters (PLAIN0); deeper levels carry run keys.
// ---------------------------------------------------------------------------------------------

// Head bookkeeping on two 64-bit lane masks (slots 0..63 in M0, 64..127 in M1).
// highest set slot <= `slot`, or -1
__device__ __forceinline__ int mask_prev_set(uint64_t M0, uint64_t M1, int slot)
{
    if (slot >= 64) {
        uint64_t m = M1 & (~0ull >> (127 - slot));
        if (m) return 127 - __clzll(m);
        return M0 ? 63 - __clzll(M0) : -1;
    }
    uint64_t m = M0 & (~0ull >> (63 - slot));
    return m ? 63 - __clzll(m) : -1;
}
// lowest set slot > `slot`, or `none`
__device__ __forceinline__ int mask_next_set(uint64_t M0, uint64_t M1, int slot, int none)
{
    if (slot < 64) {
        uint64_t m = slot == 63 ? 0ull : (M0 & (~0ull << (slot + 1)));
        if (m) return __builtin_ctzll(m);
        return M1 ? 64 + __builtin_ctzll(M1) : none;
    }
    uint64_t m = slot == 127 ? 0ull : (M1 & (~0ull << (slot - 63)));
    return m ? 64 + __builtin_ctzll(m) : none;
}

#ifdef SUFR_HIP_PROBES
__device__ unsigned long long g_walk_stats[4];       // pair walks, their characters, the longest, (spare)
#endif

// common prefix of suffixes a and b (positions in the text), 64 characters per round trip: find_lcp's byte walk
// (sufr_builder.rs:319-329) as a word walk
// A walk is blind to runs: two suffixes that continue with megabases of `N` are better served by a run key, which
// crosses a run in one step.  In-kernel walks therefore stop after WALK_CAP characters; a pair / group that has not
// parted by then stays tied, WALK_CAP characters deeper, and is keyed with a run key next.
static constexpr uint32_t WALK_CAP = 4096;

__device__ __forceinline__ uint64_t walk_lcp(const uint8_t* __restrict__ text, uint64_t n, uint64_t a, uint64_t b,
                                             uint64_t cap = ~0ull)
{
    uint64_t lim = n - (a > b ? a : b);              // characters both suffixes have
    if (lim > cap) lim = cap;
    uint64_t k = 0;
    while (k + 64 <= lim) {
        uint64_t x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = load_u64_unaligned(text + a + k + 8 * u) ^ load_u64_unaligned(text + b + k + 8 * u);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (x[u]) return k + 8 * u + (uint64_t)(__builtin_ctzll(x[u]) >> 3);
        k += 64;
    }
    while (k < lim && text[a + k] == text[b + k]) k++;
    return k;
}

// The same walk on the bit-packed code stream (alphabets of <= 4 bits per symbol): 42 symbols of a 3-bit alphabet per step, 24
// bytes -- one sector per suffix, usually the one its run key was just read from -- where the byte text costs two to four.
// Codes are the ranks of the bytes and 0 past the end of the text: equal codes <=> equal bytes, code order = byte order.
// `ca`, `cb`: the codes at the mismatch (valid when the result is below the limit).
__device__ __forceinline__ uint64_t walk_lcp_packed(const uint8_t* __restrict__ packed, int bits, uint64_t n, uint64_t a, uint64_t b,
                                                    uint64_t cap, uint32_t& ca, uint32_t& cb)
{
    uint64_t lim = n - (a > b ? a : b);
    if (lim > cap) lim = cap;
    const uint32_t K = div_by_bits(64u, bits);                 // whole codes per 64-bit word
    uint64_t k = 0;
    ca = cb = 0;
    while (k < lim) {
        const uint64_t oa = (a + k) * (uint64_t)bits, ob = (b + k) * (uint64_t)bits;
        const uint8_t* pa = packed + (oa >> 3);
        const uint8_t* pb = packed + (ob >> 3);
        uint64_t wa[3], wb[3];                                  // 24 bytes each: two words of K codes behind a bit offset below 8
#pragma unroll
        for (int u = 0; u < 3; u++) { wa[u] = __builtin_bswap64(load_u64_unaligned(pa + 8 * u)); wb[u] = __builtin_bswap64(load_u64_unaligned(pb + 8 * u)); }
        const uint32_t sa = (uint32_t)(oa & 7u), sb = (uint32_t)(ob & 7u);
        // two words of K codes, then what the 24 bytes still hold behind them: 192 - 7 - 2 K bits >= K3 whole codes (round 5: the
        // third piece was read and not looked at; 42 -> 61 symbols per round trip for 3-bit codes)
        // (K3 <= K: for 5-bit codes the division gives 13 > K = 12 -- a negative shift below; the packed path stops at 4 bits
        // today, sufr_launch.inc, but the walk must not depend on that: advisor r5)
        const uint32_t K3raw = div_by_bits(192u - 7u - 2u * K * (uint32_t)bits, bits);
        const uint32_t K3 = K3raw < K ? K3raw : K;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t qa = sa + (uint32_t)j * K * (uint32_t)bits, qb = sb + (uint32_t)j * K * (uint32_t)bits;   // < 8 + 128
            const uint32_t ra = qa & 63u, rb = qb & 63u;
            // (selects, not indexed registers)
            const uint64_t ha = qa < 64u ? wa[0] : (qa < 128u ? wa[1] : wa[2]), la = qa < 64u ? wa[1] : (qa < 128u ? wa[2] : 0ull);
            const uint64_t hb = qb < 64u ? wb[0] : (qb < 128u ? wb[1] : wb[2]), lb = qb < 64u ? wb[1] : (qb < 128u ? wb[2] : 0ull);
            const uint64_t va = ra ? ((ha << ra) | (la >> (64u - ra))) : ha;
            const uint64_t vb = rb ? ((hb << rb) | (lb >> (64u - rb))) : hb;
            const uint32_t Kj = j < 2 ? K : K3;                                   // whole codes compared in this piece
            const int sparej = 64 - (int)Kj * bits;
            if (Kj == 0u) break;
            const uint64_t x = (va ^ vb) >> sparej;
            if (x) {
                const uint32_t c = div_by_bits((uint32_t)__builtin_clzll(x << sparej), bits);
                k += c;
                if (k >= lim) return lim;
                const int sh = 64 - (int)(c + 1u) * bits;
                ca = (uint32_t)(va >> sh) & ((1u << bits) - 1u);
                cb = (uint32_t)(vb >> sh) & ((1u << bits) - 1u);
                return k;
            }
            k += Kj;
            if (k >= lim) return lim;
        }
    }
    return lim;
}

// Walk keys: a member of a tie group keyed by its common prefix L with the group's FIRST member (the reference),
// beyond the depth all members share.  A member that leaves the reference's text earlier than another is smaller than
// it iff its character there is below the reference's: {below the reference, L ascending} < reference < {above, L
// descending} -- the order of run tokens (sufr_runkey.h) with the reference in the place of the periodic extension.
// Two members part after min(L, L') characters; equal keys share L characters.
// The reference's own key stands for "agrees with the reference for WALK_CAP characters": members that do take it too
// and stay tied with it, WALK_CAP characters deeper.
static constexpr uint64_t WALK_MAX = (1ull << 62) - 1;
__device__ __forceinline__ uint64_t walk_key(bool is_ref, bool below, uint64_t L)
{
    return is_ref ? (1ull << 62) : (below ? L : ((2ull << 62) | (WALK_MAX - L)));
}
__device__ __forceinline__ uint64_t walk_key_len(uint64_t k)
{
    const uint32_t cls = (uint32_t)(k >> 62);
    return cls == 0 ? k : (cls == 1 ? (uint64_t)WALK_CAP : WALK_MAX - (k & WALK_MAX));
}
__device__ __forceinline__ uint32_t walk_key_common(uint64_t a, uint64_t b)
{
    const uint64_t la = walk_key_len(a), lb = walk_key_len(b);
    return (uint32_t)(la < lb ? la : lb);
}

// TIES_OUT (top level): records that still tie after the first ranking (equal whole key) are not iterated
// on here -- almost every window has a few of them, and re-keying them in place would keep 128 lanes busy
// for the sake of two or three.  Instead the window's tie runs are written back in sorted order and
// described by four 64-bit lane masks (members / run heads); k_build_ties packs them into a dense level
// whose finisher then works with all lanes active.
// PLANNED: the windows come from k_plan_windows: wave i takes the records [wl_head[i], wl_head[i] + wl_flag[i]) (at most 128),
// a window that starts and ends at group heads -- every group in it is finished here; the groups above a window are already
// on the next level's list.  (Rounds 2-3: fixed windows of 128, groups crossing their end re-done by a second launch.)
template <bool DEEP, bool TIES_OUT, bool PLANNED>
__global__ void __launch_bounds__(256, 5)       // 5 waves per SIMD (96 registers, no spill): the rounds are chains of random reads
k_finish(const uint64_t* __restrict__ keys, const uint32_t* idxs,
         const uint32_t* __restrict__ segs, const uint32_t* __restrict__ opos,
         const uint32_t* __restrict__ segdepth, uint32_t m,
         const uint8_t* __restrict__ text, uint64_t n, RunTable R,
         const uint16_t* __restrict__ glut, KeyParams kp, int sorted_bits,
         uint32_t* __restrict__ SA, uint32_t* __restrict__ LCP,
         uint32_t* __restrict__ wl_flag, uint32_t* __restrict__ wl_head,
         uint32_t* idx_writeback, unsigned long long* __restrict__ wmask, uint32_t* __restrict__ wcnt,
         uint32_t ncross)
{
#ifdef SUFR_HIP_PROBES
    const uint32_t popts = ncross >> 29;            // timing probes: 1 = no pair walk, 2 = no walk keys, 4 = count the walks
    ncross &= 0x1fffffffu;
#else
    constexpr uint32_t popts = 0;
#endif
    __shared__ uint64_t sh_key[4][128];
    __shared__ uint32_t sh_idx[4][128];
    __shared__ uint32_t sh_gid[4][128];
    __shared__ uint16_t s_lut[256];
    for (int i = threadIdx.x; i < 256; i += 256) s_lut[i] = glut[i];
    __syncthreads();

    // one wave per window: PLANNED -- the i-th planned window; else 128 consecutive records, and a group that crosses the
    // window's end is "large"
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (PLANNED ? (wave >= ncross) : ((uint64_t)wave * 128 >= m)) return;
    uint64_t base64 = PLANNED ? (uint64_t)wl_head[wave] : (uint64_t)wave * 128;
    const uint32_t wcount = PLANNED ? wl_flag[wave] : 128u;
    const int wv = threadIdx.x >> 6;
    auto sk = SUFR_LDS_VOLATILE(uint64_t, sh_key[wv]);
    auto si = SUFR_LDS_VOLATILE(uint32_t, sh_idx[wv]);
    auto sg = SUFR_LDS_VOLATILE(uint32_t, sh_gid[wv]);
    const int ln = (int)lane_id();
    const bool whole = sorted_bits == 0;            // groups are whole segments (no key bits sorted yet)
    const int group_shift = whole ? 0 : 64 - sorted_bits;
    // A wave owns the groups that START in its 128 records.  It works on the 128 records from its first group head on:
    // what it skips at the front belongs to the wave before, and the group that ran past its end now does so only if it
    // overhangs by more than that (a level of tie runs of ~12 records: half the windows flagged a crossing group for the
    // second pass when every wave sat on its own 128 records).
    uint32_t own_limit = 128u;                      // groups whose head slot is below this are this wave's
    if (DEEP && !TIES_OUT && !PLANNED) {
        const uint32_t b0 = (uint32_t)base64;
        const uint32_t a0 = b0 + (uint32_t)ln, a1 = b0 + 64u + (uint32_t)ln;
        const bool v0 = a0 < m, v1 = a1 < m;
        const uint64_t pk0 = v0 ? keys[a0] : ~0ull, pk1 = v1 ? keys[a1] : ~0ull;
        const uint32_t ps_0 = v0 ? segs[a0] : 0xffffffffu, ps_1 = v1 ? segs[a1] : 0xffffffffu;
        uint64_t q0 = shfl64_up1(pk0), q1 = shfl64_up1(pk1);
        uint32_t r0 = __shfl_up(ps_0, 1, WAVE), r1 = __shfl_up(ps_1, 1, WAVE);
        const uint64_t l0k = shfl64(pk0, 63); const uint32_t l0s = __shfl(ps_0, 63, WAVE);
        if (ln == 0) {
            q1 = l0k; r1 = l0s;
            if (b0 > 0) { q0 = keys[b0 - 1]; r0 = segs[b0 - 1]; }
        }
        const bool e0 = !v0 || (ln == 0 && b0 == 0) || r0 != ps_0 || (!whole && (q0 >> group_shift) != (pk0 >> group_shift));
        const bool e1 = !v1 || r1 != ps_1 || (!whole && (q1 >> group_shift) != (pk1 >> group_shift));
        const uint64_t E0 = __ballot(e0), E1 = __ballot(e1);
        const uint32_t first = E0 ? (uint32_t)__builtin_ctzll(E0) : (E1 ? 64u + (uint32_t)__builtin_ctzll(E1) : 128u);
        if (first >= 128u || (uint64_t)b0 + first >= m) {       // no group starts here: nothing to own
            if (ln == 0) wl_flag[wave] = 0u;
            return;
        }
        base64 += first;
        own_limit = 128u - first;
    }
    const uint32_t base = (uint32_t)base64;
    const uint32_t j0 = base + ln, j1 = base + 64 + ln;
    const bool in0 = j0 < m && (uint32_t)ln < wcount, in1 = j1 < m && 64u + (uint32_t)ln < wcount;

    uint64_t k0 = in0 ? keys[j0] : ~0ull, k1 = in1 ? keys[j1] : ~0ull;
    uint32_t i0 = in0 ? idxs[j0] : 0u, i1 = in1 ? idxs[j1] : 0u;
    uint32_t sg0 = 0, sg1 = 0;
    if (DEEP) { sg0 = in0 ? segs[j0] : 0xffffffffu; sg1 = in1 ? segs[j1] : 0xffffffffu; }
    uint32_t dd0 = 0, dd1 = 0;       // depth at which the record's key was taken (per segment below level 0)
    if (DEEP) { dd0 = in0 ? segdepth[sg0] : 0u; dd1 = in1 ? segdepth[sg1] : 0u; }

    // record before the window (slot -1) and after it (slot 128)
    uint64_t kprev = 0, knext = 0; uint32_t sprev = 0, snext = 0;
    const bool has_prev = base > 0;
    const bool has_next = !PLANNED && (base64 + 128) < m;     // (a planned window ends at a head)
    if (has_prev) { kprev = keys[base - 1]; if (DEEP) sprev = segs[base - 1]; }
    if (has_next) { knext = keys[base + 128]; if (DEEP) snext = segs[base + 128]; }

    // ---- head flags (group = equal segment and equal sorted bits) -------------------------------
    uint64_t p0 = shfl64_up1(k0), p1 = shfl64_up1(k1);
    uint32_t ps0 = __shfl_up(sg0, 1, WAVE), ps1 = __shfl_up(sg1, 1, WAVE);
    uint64_t last0k = shfl64(k0, 63); uint32_t last0s = __shfl(sg0, 63, WAVE);
    if (ln == 0) { p0 = kprev; ps0 = sprev; p1 = last0k; ps1 = last0s; }
    const bool segdiff0 = DEEP && (ps0 != sg0), segdiff1 = DEEP && (ps1 != sg1);
    const bool h0 = !in0 || (ln == 0 && !has_prev) || segdiff0 ||
                    (!whole && (p0 >> group_shift) != (k0 >> group_shift));
    const bool h1 = !in1 || segdiff1 || (!whole && (p1 >> group_shift) != (k1 >> group_shift));
    uint64_t last1k = shfl64(k1, 63); uint32_t last1s = __shfl(sg1, 63, WAVE);
    const bool h128 = !has_next || (DEEP && snext != last1s) ||
                      (!whole && (knext >> group_shift) != (last1k >> group_shift));
    const uint64_t H0 = __ballot(h0), H1 = __ballot(h1);

    // first record of a segment (or of everything): its LCP belongs to the parent level / is 0
    const bool first0 = (ln == 0 && !has_prev) || segdiff0;
    const bool first1 = segdiff1;
    // LCP of a head with the record before it (same segment, different group => keys differ)
    uint32_t lcp0 = 0, lcp1 = 0;
    if (!first0 && in0) lcp0 = dd0 + (DEEP ? run_key_common(p0, k0, kp.b) : plain_key_common(p0, k0, kp.b, kp.K));
    if (!first1 && in1) lcp1 = dd1 + (DEEP ? run_key_common(p1, k1, kp.b) : plain_key_common(p1, k1, kp.b, kp.K));

    // ---- groups: head slot of each slot's group, and the next head after it -----------------------
    const int g0 = mask_prev_set(H0, H1, ln), g1 = mask_prev_set(H0, H1, 64 + ln);
    const int none = h128 ? 128 : 1000;
    const int nx0 = mask_next_set(H0, H1, ln, none), nx1 = mask_next_set(H0, H1, 64 + ln, none);
    // group starts inside this window (PLANNED: all of them do)
    const bool own0 = in0 && (PLANNED || (g0 >= 0 && (uint32_t)g0 < own_limit));
    const bool own1 = in1 && (PLANNED || (g1 >= 0 && (uint32_t)g1 < own_limit));
    const bool small0 = own0 && nx0 <= 128, small1 = own1 && nx1 <= 128;
    // Only the window's last group can run past its end, so a window reports at most one large group:
    // a flag and a head position per window, compacted afterwards by a scan (k_compact_large).  (Appending
    // with one global atomic counter serialises: a quarter of all windows end in a crossing group.)
    const bool lg0 = own0 && h0 && !small0, lg1 = own1 && h1 && !small1;
    if (!PLANNED && lg0) {
        wl_head[wave] = j0;
        if (!first0) LCP[DEEP ? opos[j0] : j0] = lcp0;
        else if (!DEEP) LCP[j0] = 0;
    }
    if (!PLANNED && lg1) {
        wl_head[wave] = j1;
        if (!first1) LCP[DEEP ? opos[j1] : j1] = lcp1;
    }
    if (!PLANNED) {
        const uint64_t any_large = __ballot(lg0 || lg1);
        if (ln == 0) wl_flag[wave] = any_large ? 1u : 0u;
    }
    bool act0 = small0 && !(h0 && nx0 == ln + 1);     // member of a small group with > 1 records
    bool act1 = small1 && !(h1 && nx1 == 64 + ln + 1);
    uint32_t gid0 = act0 ? (uint32_t)g0 : (0x10000u | (uint32_t)ln);
    uint32_t gid1 = act1 ? (uint32_t)g1 : (0x10000u | (uint32_t)(64 + ln));
    uint32_t ge0 = act0 ? (uint32_t)nx0 : 0u, ge1 = act1 ? (uint32_t)nx1 : 0u;      // end of the slot's group (active slots)

    int ktype = DEEP ? 1 : 0;                 // keys currently held: 0 plain packed characters, 1 run keys, 2 walk keys
    uint32_t extra0 = 0, extra1 = 0;          // characters a capped pair walk has matched beyond the key
    const uint64_t max_round = n + 8;        // distinct suffixes separate within n characters
    bool lt0 = false, lt1 = false;           // tie flags of the last ranking
    SUFR_STAMP_DECL(8)
    for (uint64_t round = 0; __ballot(act0 || act1) != 0ull && round < max_round; round++) {
        SUFR_STAMP(ph__, 0)
#ifdef SUFR_HIP_PROBES
        ph__[5] += 1000;                      // (rounds, in thousands of a 'cycle')
#endif
        // ---- rank every active record inside its group (counting sort through LDS) ----------------
        // The group's slots are [gid, ge): the bounds are known from the head / tie masks, so the loop reads four keys per trip
        // with nothing between the loads (round 5: the loop used to find the group's end by reading a slot's group id before its
        // key -- two dependent LDS round trips per member, 44 % of the kernel's time on groups of ~27).
        sk[ln] = k0; sk[64 + ln] = k1;
        uint32_t np0 = (uint32_t)ln, np1 = (uint32_t)(64 + ln);
        auto rank_in = [&](uint64_t kme, uint32_t me, uint32_t g, uint32_t e) -> uint32_t {
            uint32_t cnt = 0, t = g;
            for (; t + 4u <= e; t += 4u) {
                const uint64_t a = sk[t], b = sk[t + 1], c = sk[t + 2], d = sk[t + 3];
                cnt += (a < kme || (a == kme && t < me)) ? 1u : 0u;
                cnt += (b < kme || (b == kme && t + 1u < me)) ? 1u : 0u;
                cnt += (c < kme || (c == kme && t + 2u < me)) ? 1u : 0u;
                cnt += (d < kme || (d == kme && t + 3u < me)) ? 1u : 0u;
            }
            for (; t < e; t++) {
                const uint64_t a = sk[t];
                cnt += (a < kme || (a == kme && t < me)) ? 1u : 0u;
            }
            return cnt;
        };
        if (act0) np0 = gid0 + rank_in(k0, (uint32_t)ln, gid0, ge0);
        if (act1) np1 = gid1 + rank_in(k1, (uint32_t)(64 + ln), gid1, ge1);
        // every lane has finished reading (one wave, in-order LDS); move the records
        if (act0) { sk[np0] = k0; si[np0] = i0; }
        if (act1) { sk[np1] = k1; si[np1] = i1; }
        if (act0) { k0 = sk[ln]; i0 = si[ln]; }
        if (act1) { k1 = sk[64 + ln]; i1 = si[64 + ln]; }
        // ---- split: slot j starts a new group if its key differs from slot j-1 of its group -------
        uint64_t q0 = (act0 && ln > 0) ? sk[ln - 1] : 0ull;
        uint64_t q1 = act1 ? sk[63 + ln] : 0ull;
        const bool same0 = act0 && (uint32_t)ln > gid0;
        const bool same1 = act1 && (uint32_t)(64 + ln) > gid1;
        const bool tie0 = same0 && q0 == k0, tie1 = same1 && q1 == k1;
        auto common = [&](uint64_t a, uint64_t b2) -> uint32_t {
            return ktype == 0 ? plain_key_common(a, b2, kp.b, kp.K) : (ktype == 1 ? run_key_common(a, b2, kp.b) : walk_key_common(a, b2));
        };
        auto advance = [&](uint64_t k) -> uint32_t {
            return ktype == 0 ? (uint32_t)kp.K : (ktype == 1 ? run_key_advance(k, 64, kp.b) : (uint32_t)walk_key_len(k));
        };
        if (same0 && !tie0) lcp0 = dd0 + common(q0, k0);
        if (same1 && !tie1) lcp1 = dd1 + common(q1, k1);
        const uint64_t T0 = __ballot(tie0), T1 = __ballot(tie1);
        // new group of a slot = nearest slot at or before it that does not tie with its predecessor
        const int ng0 = mask_prev_set(~T0, ~T1, ln), ng1 = mask_prev_set(~T0, ~T1, 64 + ln);
        const bool s0n = ln < 63 ? ((T0 >> (ln + 1)) & 1ull) : (T1 & 1ull);   // my successor ties with me
        const bool s1n = ln < 63 ? ((T1 >> (ln + 1)) & 1ull) : false;
        act0 = act0 && (tie0 || s0n);
        act1 = act1 && (tie1 || s1n);
        gid0 = act0 ? (uint32_t)ng0 : (0x10000u | (uint32_t)ln);
        gid1 = act1 ? (uint32_t)ng1 : (0x10000u | (uint32_t)(64 + ln));
        ge0 = (uint32_t)mask_next_set(~T0, ~T1, ln, 128);                  // the next slot that does not tie with its predecessor
        ge1 = (uint32_t)mask_next_set(~T0, ~T1, 64 + ln, 128);
        lt0 = tie0; lt1 = tie1;
        if (TIES_OUT) break;                  // tie runs go to the dense tie level (k_build_ties)
        SUFR_STAMP(ph__, 1)
        // ---- pairs: two suffixes that still tie (and nobody else with them) are compared directly, eight
        // characters per step, instead of being re-keyed ~20 characters per round: an exact duplicate of 100 kb is
        // 10^5 such pairs with common prefixes of up to 10^5 characters (find_lcp's byte walk, sufr_builder.rs:
        // 319-329, as a word walk).  The head lane of a pair does the walk; results travel through the LDS.
        // (the pair heads / seconds as 128-bit masks out of the tie masks -- wave-uniform words, scalar shifts -- so that a round
        // without pairs leaves here on a scalar test: round 5; each lane used to test five bits of T around its slots first)
        const uint64_t TN0 = (T0 >> 1) | (T1 << 63), TN1 = T1 >> 1;                 // tie of the slot after
        const uint64_t TNN0 = (T0 >> 2) | (T1 << 62), TNN1 = T1 >> 2;               // ... two after
        const uint64_t TP0 = T0 << 1, TP1 = (T1 << 1) | (T0 >> 63);                 // ... before
        const uint64_t PH0 = ~T0 & TN0 & ~TNN0, PH1 = ~T1 & TN1 & ~TNN1;            // heads a group of exactly two
        const uint64_t PS0 = T0 & ~TN0 & ~TP0, PS1 = T1 & ~TN1 & ~TP1;              // its second member
        if (!(popts & 1u) && (PH0 | PH1) != 0ull) {
            const bool ph0 = act0 && ((PH0 >> ln) & 1ull) != 0ull;
            const bool ph1 = act1 && ((PH1 >> ln) & 1ull) != 0ull;
            const bool ps0 = act0 && ((PS0 >> ln) & 1ull) != 0ull;
            const bool ps1 = act1 && ((PS1 >> ln) & 1ull) != 0ull;
            if (__ballot(ph0 || ph1) != 0ull) {
                auto walk = [&](uint32_t slot, uint32_t mine, uint32_t dnew) {
                    const uint32_t other = si[slot + 1];
                    const uint64_t a = (uint64_t)mine + dnew, b = (uint64_t)other + dnew;
                    const uint64_t lim = n - (a > b ? a : b);        // characters both suffixes have
                    uint32_t ca = 0, cb = 0;
                    const uint64_t k = kp.packed ? walk_lcp_packed(kp.packed, kp.b, n, a, b, WALK_CAP, ca, cb)
                                                 : walk_lcp(text, n, a, b, WALK_CAP);
                    if (k == WALK_CAP && k < lim) {                  // not parted yet: stays a pair, WALK_CAP deeper
                        sg[slot] = 0xffffffffu; sg[slot + 1] = 0xffffffffu;
                        return;
                    }
                    // the suffix that ends first (a proper prefix of the other) sorts first
                    const bool mine_first = k == lim ? a > b : (kp.packed ? ca < cb : text[a + k] < text[b + k]);
                    si[slot] = mine_first ? mine : other;
                    si[slot + 1] = mine_first ? other : mine;
                    sg[slot] = 0u;
                    sg[slot + 1] = dnew + (uint32_t)k;
#ifdef SUFR_HIP_PROBES
                    if (popts & 4u) {
                        atomicAdd(&g_walk_stats[0], 1ull); atomicAdd(&g_walk_stats[1], (unsigned long long)k);
                        atomicMax(&g_walk_stats[2], (unsigned long long)k);
                    }
#endif
                };
                if (ph0) walk((uint32_t)ln, i0, dd0 + advance(k0));
                if (ph1) walk((uint32_t)(64 + ln), i1, dd1 + advance(k1));
                if (ph0 || ps0) {
                    const uint32_t v = sg[ln];
                    if (v == 0xffffffffu) extra0 = WALK_CAP;
                    else { i0 = si[ln]; if (ps0) lcp0 = v; act0 = false; gid0 = 0x10000u | (uint32_t)ln; lt0 = false; }
                }
                if (ph1 || ps1) {
                    const uint32_t v = sg[64 + ln];
                    if (v == 0xffffffffu) extra1 = WALK_CAP;
                    else { i1 = si[64 + ln]; if (ps1) lcp1 = v; act1 = false; gid1 = 0x10000u | (uint32_t)(64 + ln); lt1 = false; }
                }
            }
        }
        SUFR_STAMP(ph__, 2)
        // ---- still tied: the whole key matched; re-key where the common prefix now ends -------------
        // Run keys and walk keys take turns: a run key looks ~20 characters ahead (or across a whole run); if that
        // did not part the group, its members are copies of something long -- the next key is the common prefix with
        // the group's first member, found by a word walk, and the round after it looks at the characters behind it.
        const bool walk_next = ktype == 1 && round >= 1 && !(popts & 2u) && __ballot(extra0 | extra1) == 0ull;
        if (walk_next) {
            // the reference: the member in the group's first slot (slots hold the sorted records by now)
            auto wkey = [&](uint32_t mine, uint32_t ref, uint32_t d) -> uint64_t {
                const uint64_t a = (uint64_t)mine + d, r2 = (uint64_t)ref + d;
                const uint64_t lim = n - (a > r2 ? a : r2);
                uint32_t ca = 0, cb = 0;
                const uint64_t L = kp.packed ? walk_lcp_packed(kp.packed, kp.b, n, a, r2, WALK_CAP, ca, cb)
                                             : walk_lcp(text, n, a, r2, WALK_CAP);
                if (L == WALK_CAP && L < lim) return walk_key(true, false, 0);      // agrees with the reference so far
                const bool below = L == lim ? a > r2 : (kp.packed ? ca < cb : text[a + L] < text[r2 + L]);
                return walk_key(false, below, L);
            };
            if (act0) {
                dd0 += advance(k0);
                const uint32_t ref = si[gid0];
                k0 = gid0 == (uint32_t)ln ? walk_key(true, false, 0) : wkey(i0, ref, dd0);
            }
            if (act1) {
                dd1 += advance(k1);
                const uint32_t ref = si[gid1];
                k1 = gid1 == (uint32_t)(64 + ln) ? walk_key(true, false, 0) : wkey(i1, ref, dd1);
            }
            ktype = 2;
        } else {
            if (act0) { dd0 += advance(k0) + extra0; extra0 = 0; }
            if (act1) { dd1 += advance(k1) + extra1; extra1 = 0; }
            if (kp.packed) {
                // both slots' sectors are asked for before either is waited for: one random-read latency per round, not two
                PackedPair f0 = {0ull, 0ull}, f1 = {0ull, 0ull};
                if (act0) f0 = packed_fetch16(kp.packed, kp.b, (uint64_t)i0 + dd0 - 1u);
                if (act1) f1 = packed_fetch16(kp.packed, kp.b, (uint64_t)i1 + dd1 - 1u);
                if (act0) k0 = make_run_key_packed(n, R, kp.b, (uint64_t)i0 + dd0, kp.packed, f0);
                if (act1) k1 = make_run_key_packed(n, R, kp.b, (uint64_t)i1 + dd1, kp.packed, f1);
            } else {
                if (act0) k0 = make_run_key(text, n, R, s_lut, kp.b, (uint64_t)i0 + dd0, 1u, nullptr);
                if (act1) k1 = make_run_key(text, n, R, s_lut, kp.b, (uint64_t)i1 + dd1, 1u, nullptr);
            }
            ktype = 1;
        }
        SUFR_STAMP(ph__, 3)
    }
    SUFR_STAMP(ph__, 0)
    if (TIES_OUT) {
        // describe this window's tie runs: members (M) and run heads (H); write the members back in order
        const uint64_t M0 = __ballot(act0), M1 = __ballot(act1);
        const uint64_t Hd0 = __ballot(act0 && !lt0), Hd1 = __ballot(act1 && !lt1);
        if (ln == 0) {
            wmask[(size_t)wave * 4 + 0] = M0; wmask[(size_t)wave * 4 + 1] = M1;
            wmask[(size_t)wave * 4 + 2] = Hd0; wmask[(size_t)wave * 4 + 3] = Hd1;
            const uint32_t nwin = (m + 127u) / 128u;      // wcnt = [records per window | runs per window]
            wcnt[wave] = (uint32_t)(__popcll(M0) + __popcll(M1));
            wcnt[(size_t)nwin + wave] = (uint32_t)(__popcll(Hd0) + __popcll(Hd1));
        }
        if (act0) idx_writeback[j0] = i0;
        if (act1) idx_writeback[j1] = i1;
    }
    // ---- write the owned, completely ordered slots ------------------------------------------------
    // (a tie-run member's SA entry, and its LCP unless it heads the run, come from the tie level)
    if (small0) {
        const bool tied = TIES_OUT && act0;
        uint32_t o = DEEP ? opos[j0] : j0;
        if (!tied) SA[o] = i0;
        if (!tied || !lt0) {
            if (!first0) LCP[o] = lcp0;
            else if (!DEEP) LCP[o] = 0;
        }
    }
    if (small1) {
        const bool tied = TIES_OUT && act1;
        uint32_t o = DEEP ? opos[j1] : j1;
        if (!tied) SA[o] = i1;
        if ((!tied || !lt1) && !first1) LCP[o] = lcp1;
    }
    SUFR_STAMP(ph__, 4)
    if (DEEP && !TIES_OUT) { SUFR_STAMP_FLUSH(2, 8) }
}

// ---------------------------------------------------------------------------------------------
// Windows for the finisher, cut at group heads (round 4).  Until then wave t sat on records [128 t, 128 t + 128): a group that
// crossed the end of its window (one window in four) was re-done by a second launch that restarted at its head (CROSS), and
// every wave first searched its records for its first group head.  Now one wave per chunk of PLAN_CH records reads the chunk's
// keys once, takes the group heads as 34 ballot words and walks them: from a head `pos`, the window runs to the LAST head within
// the next 128 records -- every group that starts in it ends in it --; if there is none the group at `pos` is larger than a
// window: its head goes to the list of the next level (with its LCP against the record before it, which the finisher used to
// write) and the walk continues at the head after it.  A chunk owns the windows and large groups that START in it; its last
// window may reach into the next chunk (two more words of heads are read for that).
// Two consecutive windows cover more than 128 records and a large group more than 128, so a chunk starts at most
// 2 * PLAN_CH / 129 + 2 = 33 windows and PLAN_CH / 129 + 1 = 16 large groups (windows next to large groups may be short, but
// then the large group's 129 records are spent): PLAN_W / PLAN_L slots per chunk, compacted by k_plan_compact after a scan.
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t PLAN_CH = 2048, PLAN_W = 48, PLAN_L = 24;

__global__ void __launch_bounds__(256)
k_plan_windows(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ segs, const uint32_t* __restrict__ opos,
               const uint32_t* __restrict__ segdepth, uint32_t m, int sorted_bits, int bits,
               uint32_t* __restrict__ LCP, uint2* __restrict__ wslots, uint32_t* __restrict__ lslots,
               uint32_t* __restrict__ wcnt, uint32_t* __restrict__ lcnt, uint32_t nchunk)
{
    const uint32_t chunk = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (chunk >= nchunk) return;
    const uint32_t ln = lane_id();
    const uint64_t c0 = (uint64_t)chunk * PLAN_CH;
    const bool whole = sorted_bits == 0;
    const int shift = whole ? 0 : 64 - sorted_bits;
    // heads of [c0, c0 + 34 * 64): word `it` ends up in lane `it` of hv.  Position m counts as a head (the end of everything).
    uint64_t hv = 0;
    uint64_t lastk = 0; uint32_t lasts = 0;
    if (c0 > 0) { lastk = keys[c0 - 1]; lasts = segs[c0 - 1]; }
#pragma unroll 2
    for (uint32_t it = 0; it < 34u; it++) {
        const uint64_t j = c0 + 64ull * it + ln;
        const bool valid = j < m;
        const uint64_t k = valid ? keys[j] : 0ull;
        const uint32_t sg = valid ? segs[j] : 0xffffffffu;
        uint64_t pk = shfl64_up1(k); uint32_t ps = __shfl_up(sg, 1, WAVE);
        if (ln == 0) { pk = lastk; ps = lasts; }
        const bool head = valid ? (j == 0 || ps != sg || (!whole && (pk >> shift) != (k >> shift))) : j == m;
        const uint64_t H = __ballot(head);
        if (ln == it) hv = H;
        lastk = shfl64(k, 63); lasts = __shfl(sg, 63, WAVE);
        if (c0 + 64ull * (it + 1) > m) break;                     // (the word that holds position m is in)
    }
    auto word = [&](uint32_t w) -> uint64_t { return shfl64(hv, (int)w); };      // w uniform
    // first head at or after `from` (chunk-relative), below `lim_w` words; PLAN_CH * 2 if none
    auto next_head = [&](uint32_t from, uint32_t lim_w) -> uint32_t {
        uint32_t w = from >> 6;
        if (w >= lim_w) return 2u * PLAN_CH;
        uint64_t x = word(w) & (~0ull << (from & 63u));
        while (!x) { if (++w >= lim_w) return 2u * PLAN_CH; x = word(w); }
        return (w << 6) + (uint32_t)__builtin_ctzll(x);
    };
    uint32_t nw = 0, nl = 0;
    uint32_t pos = next_head(0, 32);
    while (pos < PLAN_CH && c0 + pos < m) {
        // the last head in (pos, pos + 128]
        const uint32_t lim = pos + 128u;
        uint32_t h = 0;
        for (int w = (int)(lim >> 6); w >= (int)(pos >> 6) && !h; w--) {
            uint64_t x = word((uint32_t)w);
            if ((uint32_t)w == (lim >> 6)) x &= (lim & 63u) == 63u ? ~0ull : ((2ull << (lim & 63u)) - 1ull);
            if ((uint32_t)w == (pos >> 6)) x &= (pos & 63u) == 63u ? 0ull : (~0ull << ((pos & 63u) + 1u));
            if (x) h = ((uint32_t)w << 6) + 63u - (uint32_t)__builtin_clzll(x);
        }
        if (h) {
            if (ln == 0 && nw < PLAN_W) wslots[(size_t)chunk * PLAN_W + nw] = make_uint2((uint32_t)(c0 + pos), h - pos);
            nw++;
            pos = h;
        } else {
            const uint32_t j = (uint32_t)(c0 + pos);
            if (ln == 0) {
                if (nl < PLAN_L) lslots[(size_t)chunk * PLAN_L + nl] = j;
                // its LCP with the record before it (same segment, another group: the keys differ)
                if (j > 0) {
                    const uint32_t sg = segs[j];
                    if (segs[j - 1] == sg) LCP[opos[j]] = segdepth[sg] + run_key_common(keys[j - 1], keys[j], bits);
                }
            }
            nl++;
            pos = next_head(lim + 1u, 32);
        }
    }
    if (ln == 0) { wcnt[chunk] = nw < PLAN_W ? nw : PLAN_W; lcnt[chunk] = nl < PLAN_L ? nl : PLAN_L; }
}

// the chunks' slots -> dense lists: windows (start, records) and the heads of the groups above a window
__global__ void __launch_bounds__(256)
k_plan_compact(const uint2* __restrict__ wslots, const uint32_t* __restrict__ lslots, const uint32_t* __restrict__ wcnt,
               const uint32_t* __restrict__ lcnt, const uint32_t* __restrict__ woff, const uint32_t* __restrict__ loff,
               uint32_t nchunk, uint32_t* __restrict__ win_start, uint32_t* __restrict__ win_count,
               uint32_t* __restrict__ heads)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t chunk = t / (PLAN_W + PLAN_L), slot = t % (PLAN_W + PLAN_L);
    if (chunk >= nchunk) return;
    if (slot < PLAN_W) {
        if (slot < wcnt[chunk]) {
            const uint2 w = wslots[(size_t)chunk * PLAN_W + slot];
            win_start[woff[chunk] + slot] = w.x; win_count[woff[chunk] + slot] = w.y;
        }
    } else if (slot - PLAN_W < lcnt[chunk])
        heads[loff[chunk] + slot - PLAN_W] = lslots[(size_t)chunk * PLAN_L + slot - PLAN_W];
}

// Dense level out of the tie runs that k_finish<false, true> described per window:
// record t = rec_off[w] + (members below the slot) takes the written-back suffix of slot base + slot, its
// segment is run_off[w] + (run heads at or below the slot) - 1, its output position is the slot itself;
// every segment is re-keyed `depth0` characters in (the whole plain key matched).
// DENSE = false: one thread per window walks its members (a window holds a couple of tied records on average: 3 per
// window on the soft-mask-ignored human stand-in, where one wave per window costs 2.5 x as much).  DENSE = true: one wave
// per window, lane l takes slots l and 64 + l -- the members' ordinals are popcounts of the masks below the lane, so the
// suffixes are read and the records written side by side (22 tied records per window with the repeats indexed: the
// thread-per-window walk, every load and store its own sector, took 16.7 ms for 502 M records).
template <bool DENSE>
__global__ void __launch_bounds__(256)
k_build_ties(const unsigned long long* __restrict__ wmask, const uint32_t* __restrict__ rec_off,
             const uint32_t* __restrict__ run_off, const uint32_t* __restrict__ idxs, uint32_t m,
             uint32_t depth0, uint32_t* __restrict__ idx_new, uint32_t* __restrict__ seg_new,
             uint32_t* __restrict__ opos_new, uint32_t* __restrict__ newdepth, uint8_t* __restrict__ newperiod)
{
    const uint32_t w = DENSE ? (blockIdx.x * 256 + threadIdx.x) >> 6 : blockIdx.x * 256 + threadIdx.x;
    const uint64_t base64 = (uint64_t)w * 128;
    if (base64 >= m) return;
    const uint32_t base = (uint32_t)base64;
    const ulonglong2 mm = *reinterpret_cast<const ulonglong2*>(wmask + (size_t)w * 4);
    if ((mm.x | mm.y) == 0ull) return;
    const ulonglong2 hh = *reinterpret_cast<const ulonglong2*>(wmask + (size_t)w * 4 + 2);
    const uint32_t t0 = rec_off[w];
    const uint32_t s0 = run_off[w] - 1u;          // + run heads at or below the slot
    if (DENSE) {
        const uint64_t M0 = mm.x, M1 = mm.y, H0 = hh.x, H1 = hh.y;
        const uint32_t ln = lane_id();
        const uint64_t lt = (1ull << ln) - 1ull, le = lt | (1ull << ln);
        if ((M0 >> ln) & 1ull) {
            const uint32_t t = t0 + (uint32_t)__popcll(M0 & lt);
            const uint32_t g = s0 + (uint32_t)__popcll(H0 & le);
            idx_new[t] = idxs[base + ln];
            seg_new[t] = g;
            opos_new[t] = base + ln;
            if ((H0 >> ln) & 1ull) { newdepth[g] = depth0; newperiod[g] = 1; }
        }
        if ((M1 >> ln) & 1ull) {
            const uint32_t t = t0 + (uint32_t)(__popcll(M0) + __popcll(M1 & lt));
            const uint32_t g = s0 + (uint32_t)(__popcll(H0) + __popcll(H1 & le));
            idx_new[t] = idxs[base + 64u + ln];
            seg_new[t] = g;
            opos_new[t] = base + 64u + ln;
            if ((H1 >> ln) & 1ull) { newdepth[g] = depth0; newperiod[g] = 1; }
        }
    } else {
        const uint64_t M[2] = {mm.x, mm.y}, H[2] = {hh.x, hh.y};
        uint32_t t = t0, sgm = s0;               // incremented at every run head
        for (int half = 0; half < 2; half++) {
            uint64_t left = M[half];
            while (left) {
                const int bit = __builtin_ctzll(left);
                left &= left - 1;
                const uint32_t slot = (uint32_t)(half * 64 + bit);
                if ((H[half] >> bit) & 1ull) {
                    sgm++;
                    newdepth[sgm] = depth0;
                    newperiod[sgm] = 1;
                }
                idx_new[t] = idxs[base + slot];
                seg_new[t] = sgm;
                opos_new[t] = base + slot;
                t++;
            }
        }
    }
}

// heads[off[w]] = head position of window w's large group (off = exclusive scan of the window flags)
__global__ void __launch_bounds__(256)
k_compact_large(const uint32_t* __restrict__ wl_flag, const uint32_t* __restrict__ wl_off,
                const uint32_t* __restrict__ wl_head, uint32_t nwin, uint32_t* __restrict__ heads)
{
    uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w < nwin && wl_flag[w]) heads[wl_off[w]] = wl_head[w];
}

// ---------------------------------------------------------------------------------------------
// Deeper levels: large groups are re-keyed where their common prefix ends.
// ---------------------------------------------------------------------------------------------
// keys[e] = run key of suffix idx[e] taken segdepth[seg[e]] characters in
__global__ void __launch_bounds__(256)
k_gather_keys(const uint8_t* __restrict__ text, uint64_t n, RunTable R,
              const uint16_t* __restrict__ glut, const uint32_t* __restrict__ idx,
              const uint32_t* __restrict__ seg, const uint32_t* __restrict__ segdepth,
              const uint8_t* __restrict__ segperiod, uint32_t m, KeyParams kp, uint64_t* __restrict__ keys,
              uint32_t* __restrict__ max_token_bits)
{
    __shared__ uint16_t s_lut[256];
    for (int i = threadIdx.x; i < 256; i += 256) s_lut[i] = glut[i];
    __syncthreads();
    uint32_t e = blockIdx.x * 256 + threadIdx.x;
    uint32_t tb = 0;
    if (e < m) {
        const uint64_t k =
            make_run_key(text, n, R, s_lut, kp.b, (uint64_t)idx[e] + segdepth[seg[e]], segperiod[seg[e]], kp.packed);
        keys[e] = k;
        tb = (uint32_t)decode_run_token(k).tokbits;
    }
    // longest run token of the level: the level must sort at least that many key bits, otherwise a group
    // whose members differ only inside their tokens would be re-keyed at the same depth for ever
    if (max_token_bits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tb = max(tb, (uint32_t)__shfl_down(tb, o, WAVE));
        if (lane_id() == 0 && tb > 34u) atomicMax(max_token_bits, tb);
    }
}

// the same for the members of the listed groups only, a workgroup per group: a level whose keys travelled with its records
// (k_build_level) re-keys just the groups that were found periodic since
__global__ void __launch_bounds__(256)
k_gather_keys_listed(const uint8_t* __restrict__ text, uint64_t n, RunTable R, const uint16_t* __restrict__ glut,
                     const uint32_t* __restrict__ idx, const uint32_t* __restrict__ list, const uint32_t* __restrict__ start,
                     const uint32_t* __restrict__ size, const uint32_t* __restrict__ segdepth,
                     const uint8_t* __restrict__ segperiod, KeyParams kp, uint64_t* __restrict__ keys)
{
    __shared__ uint16_t s_lut[256];
    for (int i = threadIdx.x; i < 256; i += 256) s_lut[i] = glut[i];
    __syncthreads();
    const uint32_t g = list[blockIdx.x];
    const uint32_t t0 = start[g], t1 = t0 + size[g];
    const uint32_t d = segdepth[g], pi = segperiod[g];
    // (gridDim.y workgroups share a group: a tandem array's tie run is 10^4-10^5 members whose periodic extensions are walked)
    for (uint32_t t = t0 + blockIdx.y * 256u + threadIdx.x; t < t1; t += 256u * gridDim.y)
        keys[t] = make_run_key(text, n, R, s_lut, kp.b, (uint64_t)idx[t] + d, pi, kp.packed);
}

// size of every large group (upper bound of (segment, sorted bits) in the sorted records) and the depth
// at which its members will be re-keyed: parent depth + characters covered by the sorted bits
template <bool DEEP>
__global__ void __launch_bounds__(256)
k_group_extent(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ segs,
               const uint32_t* __restrict__ segdepth, uint32_t m, int sorted_bits, KeyParams kp,
               const uint32_t* __restrict__ heads, const uint32_t* __restrict__ idxs,
               const uint8_t* __restrict__ text, uint32_t L, uint32_t* __restrict__ sizes,
               uint32_t* __restrict__ newdepth, uint8_t* __restrict__ newperiod, uint32_t* __restrict__ maxsize,
               uint32_t* __restrict__ periodic_list = nullptr)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= L) return;
    const bool whole = sorted_bits == 0;
    const int group_shift = whole ? 0 : 64 - sorted_bits;
    uint32_t h = heads[k];
    uint64_t hk = keys[h];
    uint64_t top = whole ? 0ull : (hk >> group_shift);
    uint32_t sg = DEEP ? segs[h] : 0u;
    uint32_t lo = h + 1, hi = m;    // first position > h that is not in the group
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        bool in_group = (!DEEP || segs[mid] == sg) && (whole || (keys[mid] >> group_shift) == top);
        if (in_group) lo = mid + 1; else hi = mid;
    }
    uint32_t sz = lo - h;
    sizes[k] = sz;
    const uint32_t nd = DEEP ? segdepth[sg] + run_key_advance(hk, sorted_bits, kp.b)
                             : (uint32_t)(sorted_bits / kp.b);
    newdepth[k] = nd;
    // smallest period (1..8) of the last 16 characters of the group's common prefix, else 1:
    // tandem arrays are then crossed in one step by the periodic run token of make_run_key
    uint32_t pi = 1;
    if (nd >= 12 && kp.detect_period) {
        const uint8_t* e = text + (uint64_t)idxs[h] + nd;    // one past the common prefix
        const uint32_t win = nd < 16 ? nd : 16;
        for (uint32_t c = 1; c <= 8 && c * 2 <= win; c++) {
            bool ok = true;
            for (uint32_t i = 1; i + c <= win && ok; i++) ok = e[-(int)i] == e[-(int)(i + c)];
            if (ok) { pi = c; break; }
        }
    }
    newperiod[k] = (uint8_t)pi;
    if (pi != 1) {                               // periodic groups of the level (rare): counted in the high half of the 8-byte scalar
        const uint32_t at = atomicAdd(maxsize + 1, 1u);
        if (periodic_list) periodic_list[at] = k;
    }
    // (only a group that would raise the value: atomics on one address are served one by one in the L2)
    if (sz > __hip_atomic_load(maxsize, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxsize, sz);
}

// new level: slot t of the new active array takes the record src = heads[k] + (t - start[k])
template <bool DEEP>
__global__ void __launch_bounds__(256)
k_build_level(const uint32_t* __restrict__ idx, uint32_t istride, const uint32_t* __restrict__ opos,
              const uint32_t* __restrict__ heads, const uint32_t* __restrict__ start, uint32_t L,
              uint32_t m_new, uint32_t* __restrict__ idx_new, uint32_t* __restrict__ seg_new,
              uint32_t* __restrict__ opos_new, const uint64_t* __restrict__ key_old = nullptr,
              uint64_t* __restrict__ key_new = nullptr)
{
    // the 256 slots of a workgroup lie in a handful of consecutive groups: two lanes bracket them over the whole table
    // (19 dependent probes at 4 * 10^5 groups), the others search the bracket
    __shared__ uint32_t s_br[2];
    const uint32_t tb = blockIdx.x * 256;
    if (threadIdx.x < 2) {
        const uint32_t q = threadIdx.x == 0 ? tb : min(tb + 255u, m_new - 1u);
        uint32_t a = 0, b = L;   // last k with start[k] <= q
        while (b - a > 1) {
            const uint32_t mid = a + (b - a) / 2;
            if (start[mid] <= q) a = mid; else b = mid;
        }
        s_br[threadIdx.x] = a;
    }
    __syncthreads();
    uint32_t t = tb + threadIdx.x;
    if (t >= m_new) return;
    uint32_t lo = s_br[0], hi = s_br[1] + 1u;   // last k with start[k] <= t
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (start[mid] <= t) lo = mid; else hi = mid;
    }
    uint32_t src = heads[lo] + (t - start[lo]);
    idx_new[t] = idx[(size_t)src * istride];     // istride 3: the index field of 12-byte records
    seg_new[t] = lo;
    opos_new[t] = DEEP ? opos[src] : src;
    // (groups that are re-keyed at the depth their keys were taken at -- the tie level's whole runs -- keep their keys)
    if (key_new) key_new[t] = key_old[src];
}

// ---------------------------------------------------------------------------------------------
// generic exclusive scan of u32 (3 kernels): used for the large-group start offsets
// ---------------------------------------------------------------------------------------------
static constexpr int SCAN_ITEMS = 2048;  // per workgroup

__global__ void __launch_bounds__(256)
k_scan_reduce(const uint32_t* __restrict__ in, uint32_t count, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t s[4];
    uint32_t b0 = blockIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
    for (uint32_t i = b0 + threadIdx.x; i < min(b0 + SCAN_ITEMS, count); i += 256) sum += in[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, WAVE);
    if (lane_id() == 0) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(256)
k_scan_partials(uint32_t* __restrict__ partial, uint32_t nblocks, unsigned long long* __restrict__ total,
                unsigned long long* __restrict__ total2 = nullptr /* a second place for the total (round 5: instead of a device-to-device copy, a launch of its own) */)
{
    // single workgroup, sequential chunks of 256
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 256) {
        uint32_t i = b0 + threadIdx.x;
        uint32_t v = i < nblocks ? partial[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            uint32_t t = __shfl_up(incl, o, WAVE);
            if ((int)lane_id() >= o) incl += t;
        }
        if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += s_w[w];
        uint32_t carry = s_carry;
        if (i < nblocks) partial[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = (unsigned long long)s_carry;
    if (threadIdx.x == 0 && total2) *total2 = (unsigned long long)s_carry;
}

__global__ void __launch_bounds__(256)
k_scan_apply(const uint32_t* __restrict__ in, uint32_t count, const uint32_t* __restrict__ partial,
             uint32_t* __restrict__ out)
{
    // one workgroup scans its SCAN_ITEMS sequentially in chunks of 256
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    uint32_t b0 = blockIdx.x * SCAN_ITEMS;
    if (threadIdx.x == 0) s_carry = partial[blockIdx.x];
    __syncthreads();
    for (uint32_t c = 0; c < SCAN_ITEMS; c += 256) {
        uint32_t i = b0 + c + threadIdx.x;
        uint32_t v = i < count ? in[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            uint32_t t = __shfl_up(incl, o, WAVE);
            if ((int)lane_id() >= o) incl += t;
        }
        if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += s_w[w];
        uint32_t carry = s_carry;
        if (i < count) out[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + wbase + incl;
        __syncthreads();
    }
}

// the same scan for short arrays (segment tables, per-tile counts of a small genome) in ONE launch: a 1024-thread
// workgroup, SCAN1_PER consecutive values per thread (three launches of ~5 us each, nine times per build, were a
// quarter of the kernel time of a 4.6 Mb genome)
static constexpr int SCAN1_PER = 16;
static constexpr uint32_t SCAN1_MAX = 1024u * SCAN1_PER;

__global__ void __launch_bounds__(1024)
k_scan_small(const uint32_t* __restrict__ in, uint32_t count, uint32_t* __restrict__ out,
             unsigned long long* __restrict__ total, unsigned long long* __restrict__ total2 = nullptr)
{
    __shared__ uint32_t s_w[16];
    const uint32_t b0 = threadIdx.x * SCAN1_PER;
    uint32_t v[SCAN1_PER];
    uint32_t local = 0;
#pragma unroll
    for (int k = 0; k < SCAN1_PER; k++) { v[k] = b0 + k < count ? in[b0 + k] : 0u; local += v[k]; }
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, WAVE);
        if ((int)lane_id() >= o) incl += t;
    }
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - local;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += s_w[w];
#pragma unroll
    for (int k = 0; k < SCAN1_PER; k++) { if (b0 + k < count) out[b0 + k] = run; run += v[k]; }
    if (threadIdx.x == 1023 && total) *total = (unsigned long long)run;
    if (threadIdx.x == 1023 && total2) *total2 = (unsigned long long)run;
}

// ... and for arrays of up to SCANM_MAX values (per-cell tables of a level of a small genome, window flags): the same workgroup
// walks the array in pieces of SCAN1_MAX with a running carry -- ~2 us per piece against two more launches of ~5 us each plus
// the gaps between dependent dispatches
static constexpr uint32_t SCANM_MAX = 16u * SCAN1_MAX;

__global__ void __launch_bounds__(1024)
k_scan_medium(const uint32_t* __restrict__ in, uint32_t count, uint32_t* __restrict__ out,
              unsigned long long* __restrict__ total, unsigned long long* __restrict__ total2 = nullptr)
{
    __shared__ uint32_t s_w[2][16];
    uint32_t carry = 0;
    int flip = 0;
    for (uint32_t base = 0; base < count; base += SCAN1_MAX, flip ^= 1) {
        const uint32_t b0 = base + threadIdx.x * SCAN1_PER;
        uint32_t v[SCAN1_PER];
        uint32_t local = 0;
        if (b0 + SCAN1_PER <= count) {
#pragma unroll
            for (int q = 0; q < SCAN1_PER / 4; q++) {
                const uint4 a = reinterpret_cast<const uint4*>(in + b0)[q];
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
#pragma unroll
            for (int k = 0; k < SCAN1_PER; k++) local += v[k];
        } else {
#pragma unroll
            for (int k = 0; k < SCAN1_PER; k++) { v[k] = b0 + k < count ? in[b0 + k] : 0u; local += v[k]; }
        }
        uint32_t incl = local;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, WAVE);
            if ((int)lane_id() >= o) incl += t;
        }
        if (lane_id() == 63) s_w[flip][threadIdx.x >> 6] = incl;
        __syncthreads();                                   // (two sets of wave sums: one barrier per piece)
        uint32_t run = carry + incl - local, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) { const uint32_t t = s_w[flip][w]; if (w < (threadIdx.x >> 6)) run += t; all += t; }
        carry += all;
        if (b0 + SCAN1_PER <= count) {
#pragma unroll
            for (int q = 0; q < SCAN1_PER / 4; q++) {
                uint4 a;
                a.x = run; run += v[4 * q]; a.y = run; run += v[4 * q + 1]; a.z = run; run += v[4 * q + 2]; a.w = run; run += v[4 * q + 3];
                reinterpret_cast<uint4*>(out + b0)[q] = a;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN1_PER; k++) { if (b0 + k < count) out[b0 + k] = run; run += v[k]; }
        }
    }
    if (threadIdx.x == 0 && total) *total = (unsigned long long)carry;
    if (threadIdx.x == 0 && total2) *total2 = (unsigned long long)carry;
}

// ---------------------------------------------------------------------------------------------
// debug / verification kernel: first index where records are not sorted by (seg, key >> shift)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_check_sorted(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ segs, uint32_t m,
               int shift, uint32_t* __restrict__ first_bad)
{
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0 || j >= m) return;
    bool bad;
    if (segs) {
        bad = segs[j - 1] > segs[j] || (segs[j - 1] == segs[j] && (keys[j - 1] >> shift) > (keys[j] >> shift));
    } else {
        bad = (keys[j - 1] >> shift) > (keys[j] >> shift);
    }
    if (bad) atomicMin(first_bad, j);
}

// ---------------------------------------------------------------------------------------------
// --max-query-len builds (sufr_builder.rs:310-314, 350-359, 668-683).  Suffixes that agree on their
// first L characters are equal for the reference's comparator; its merge then emits the larger position
// first (701-712).  The exact arrays are built first; these kernels cap the LCP at L, find the runs of
// ranks with LCP >= L, and re-order every run by descending position with the ordinary LSD passes over
// (run ordinal, n-1-position).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mql_flags(uint32_t* __restrict__ lcp, uint32_t s, uint32_t L, uint32_t* __restrict__ member)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= s) return;
    const uint32_t raw = lcp[r];
    const bool tie = r > 0 && raw >= L;                       // same first L characters as rank r-1
    const bool nxt = r + 1 < s && lcp[r + 1] >= L;            // (capped or not, the test reads the same)
    if (raw > L) lcp[r] = L;
    member[r] = (tie || nxt) ? 1u : 0u;
}

// The capped build proper (round 6; sufr_launch.inc "mql_fast": L fits the key).  Records are keyed by their first L characters
// and, below them, the top bits of the complement of their position (k_msd_part_text<.., POS>): the MSD levels and the leaf sort
// order equal L-prefixes by descending position as far as those bits tell.  What still ties is keyed by the whole complement
// (k_pos_keys: one level, no text read), and every LCP the key comparisons computed beyond L characters is cut back to L.
__global__ void __launch_bounds__(256)
k_pos_keys(const uint32_t* __restrict__ idx, uint32_t m, uint64_t* __restrict__ keys)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < m) keys[t] = (uint64_t)(~idx[t]) << 32;
}
__global__ void __launch_bounds__(256)
k_mql_cap(uint32_t* __restrict__ lcp, uint32_t s, uint32_t L)
{
    // 16 bytes per lane; the stores are skipped where nothing changes (most ranks of a long cap)
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t r = q * 4u;
    if (r + 4u <= s) {
        uint4 v = reinterpret_cast<const uint4*>(lcp)[q];
        if (v.x > L || v.y > L || v.z > L || v.w > L) {
            v.x = v.x > L ? L : v.x; v.y = v.y > L ? L : v.y; v.z = v.z > L ? L : v.z; v.w = v.w > L ? L : v.w;
            reinterpret_cast<uint4*>(lcp)[q] = v;
        }
    } else {
        for (uint32_t i = r; i < s; i++) if (lcp[i] > L) lcp[i] = L;
    }
}

__global__ void __launch_bounds__(256)
k_mql_compact(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ lcp,
              const uint32_t* __restrict__ member, const uint32_t* __restrict__ off, uint32_t s, uint32_t L,
              uint32_t* __restrict__ idx, uint32_t* __restrict__ slot, uint32_t* __restrict__ head)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= s || !member[r]) return;
    const uint32_t j = off[r];
    idx[j] = sa[r];
    slot[j] = r;
    head[j] = (r > 0 && lcp[r] >= L) ? 0u : 1u;
}

// run ordinal of every compacted record (in place of the scan of the heads), first record of every run, and the key of
// the per-run sort: last_pos - position in the top pos_bits bits (k_group_sort_* sort key bits [64 - pos_bits, 64))
__global__ void __launch_bounds__(256)
k_mql_groups(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ head, uint32_t* gid /* in: heads before, out: run */,
             uint32_t m, int pos_bits, uint32_t last_pos, uint64_t* __restrict__ key, uint32_t* __restrict__ run_start)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint32_t g = gid[j] + head[j] - 1u;
    gid[j] = g;
    key[j] = (uint64_t)(last_pos - idx[j]) << (64 - pos_bits);
    if (head[j]) run_start[g] = j;
}

// size[g] = start[g + 1] - start[g] (the last one ends at m); *maxsize = the largest
__global__ void __launch_bounds__(256)
k_sizes_from_starts(const uint32_t* __restrict__ start, uint32_t L, uint32_t m, uint32_t* __restrict__ size,
                    unsigned long long* __restrict__ maxsize)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t sz = 0;
    if (g < L) { sz = (g + 1u < L ? start[g + 1u] : m) - start[g]; size[g] = sz; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sz = max(sz, (uint32_t)__shfl_xor((int)sz, o, WAVE));
    // one per wave, and only a wave that would raise the value: atomics on ONE address are served one by one in the L2
    // (~8 ns each: 2.1 ms for the 2.6 * 10^5 waves of `-m 12` on 100 Mb)
    if (lane_id() == 0 && (unsigned long long)sz > __hip_atomic_load(maxsize, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(maxsize, (unsigned long long)sz);
}

// keys of the whole-array passes (runs of millions of records: small L on a large text): run | last_pos - position
__global__ void __launch_bounds__(256)
k_mql_keys(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ run, uint32_t m, int pos_bits, uint32_t last_pos,
           uint64_t* __restrict__ key)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    key[j] = ((uint64_t)run[j] << pos_bits) | (uint64_t)(last_pos - idx[j]);
}

__global__ void __launch_bounds__(256)
k_scatter_slots(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ slot, uint32_t m,
                uint32_t* __restrict__ sa)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < m) sa[slot[j]] = idx[j];
}

// ---------------------------------------------------------------------------------------------
// --seed-mask builds (sufr_builder.rs:272-300, 350-359, 668-683; types.rs:36-200).  A suffix is ordered
// by the characters at the mask's "care" offsets that lie inside the text; equal keys come out in
// descending position (the merge takes the shorter suffix first, 701-712).  This is not a suffix order,
// so it does not go through the text-streaming partition kernel: eligible positions are listed in
// descending order and LSD-sorted on the care characters, last block of characters first.
// ---------------------------------------------------------------------------------------------
// A shard of a seed-mask build (several GPUs): the suffixes whose first key digit -- the codes of the first `mch` care
// symbols, b bits each, first one highest; the top digit of the LSD passes of sort_masked -- lies in [lo, hi).  The
// masked order is "care symbols, then descending position": equal keys never straddle a digit boundary, so the shards
// concatenate to the one-GPU arrays exactly as the first-digit shards of a plain build do.  hi == 0: no filter.
struct MaskShard {
    const uint32_t* offs;       // care offsets (device)
    int mch, b;
    uint32_t lo, hi;
};
__device__ __forceinline__ uint32_t mask_top_digit(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* lut, uint64_t p,
                                                   const MaskShard& ms)
{
    uint32_t d = 0;
    for (int c = 0; c < ms.mch; c++) {
        const uint64_t q = p + ms.offs[c];
        d = (d << ms.b) | (q < n ? (uint32_t)(lut[text[q]] & 0x3ffu) : 0u);
    }
    return d;
}
__device__ __forceinline__ bool mask_in_shard(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* lut, uint64_t p,
                                              const MaskShard& ms)
{
    if (ms.hi == 0u) return true;
    const uint32_t d = mask_top_digit(text, n, lut, p, ms);
    return d >= ms.lo && d < ms.hi;
}

// first key digit of every eligible suffix, counted (the "pivots" of a sharded seed-mask build: exact counts, every rank the same)
__global__ void __launch_bounds__(256)
k_mask_top_hist(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut, MaskShard ms,
                uint32_t* __restrict__ hist /* 1 << (mch * b) */)
{
    __shared__ uint16_t s_lut[256];
    __shared__ uint32_t s_h[1024];
    s_lut[threadIdx.x] = glut[threadIdx.x];
    for (int i = threadIdx.x; i < 1024; i += 256) s_h[i] = 0;
    __syncthreads();
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * 16;
    for (int e = 0; e < 16; e++)
        if (p0 + e < n && (s_lut[text[p0 + e]] & 0x8000u)) atomicAdd(&s_h[mask_top_digit(text, n, s_lut, p0 + e, ms) & 1023u], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

__global__ void __launch_bounds__(256)
k_elig_count(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
             uint32_t* __restrict__ tilecnt, MaskShard ms)
{
    __shared__ uint16_t s_lut[256];
    __shared__ uint32_t s_w[4];
    s_lut[threadIdx.x] = glut[threadIdx.x];
    __syncthreads();
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * 16;
    uint32_t cnt = 0;
#pragma unroll
    for (int e = 0; e < 16; e++)
        if (p0 + e < n && (s_lut[text[p0 + e]] & 0x8000u) && mask_in_shard(text, n, s_lut, p0 + e, ms)) cnt++;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, WAVE);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(256)
k_elig_emit_desc(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
                 const uint32_t* __restrict__ tileoff, uint32_t s, uint32_t* __restrict__ idx, MaskShard ms)
{
    __shared__ uint16_t s_lut[256];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_pos[TILE];                          // the tile's eligible positions in text order
    s_lut[threadIdx.x] = glut[threadIdx.x];
    __syncthreads();
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * 16;
    uint8_t v[16];
    if (p0 + 16 <= n) *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(text + p0);
    else for (int e = 0; e < 16; e++) v[e] = p0 + e < n ? text[p0 + e] : 0;
    uint32_t el = 0, cnt = 0;
#pragma unroll
    for (int e = 0; e < 16; e++)
        if (p0 + e < n && (s_lut[v[e]] & 0x8000u) && mask_in_shard(text, n, s_lut, p0 + e, ms)) { el |= 1u << e; cnt++; }
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(incl, o, WAVE);
        if ((int)lane_id() >= o) incl += t;
    }
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t q = incl - cnt;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) q += s_w[w];
#pragma unroll
    for (int e = 0; e < 16; e++)
        if (el & (1u << e)) s_pos[q++] = (uint32_t)(p0 + e);
    __syncthreads();
    // out in DESCENDING position: consecutive lanes, consecutive (descending) places -- whole lines instead of one line per lane
    // and store (round 3's version: 14.9 ms for the 2.95 G positions of the stand-in)
    const uint32_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3], base = tileoff[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < total; j += 256) idx[s - 1u - (base + j)] = s_pos[j];
}

// key of one block of care characters: code of text[pos + offs[k]] (0 past the end), first one highest
__global__ void __launch_bounds__(256)
k_mask_keys(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
            const uint32_t* __restrict__ idx, uint32_t m, const uint32_t* __restrict__ offs, int cnt, int b,
            uint64_t* __restrict__ key)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t p = idx[j];
    uint64_t k = 0;
    for (int c = 0; c < cnt; c++) {
        const uint64_t q = p + offs[c];
        const uint64_t code = q < n ? (uint64_t)(glut[text[q]] & 0x3ffu) : 0ull;
        k |= code << (64 - b * (c + 1));
    }
    key[j] = k;
}

// The same through LDS: the first `span` symbols of the suffix arrive eight bytes per load (the workspace copy of the
// text is padded) and the care symbols are picked out of the thread's own row -- one visit of the text per suffix
// instead of one byte load per care symbol, each of them 64 sectors at random per wave instruction.
__global__ void __launch_bounds__(256)
k_mask_keys_staged(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ glut,
                   const uint32_t* __restrict__ idx, uint32_t m, const uint32_t* __restrict__ offs, int cnt, int b,
                   uint32_t span, uint64_t* __restrict__ key)
{
    extern __shared__ __align__(16) uint8_t s_rows[];      // 256 rows of `pitch` bytes
    __shared__ uint32_t s_offs[64];
    __shared__ uint16_t s_lut[256];
    const uint32_t words = (span + 7u) / 8u, pitch = words * 8u;
    if ((int)threadIdx.x < cnt) s_offs[threadIdx.x] = offs[threadIdx.x];
    s_lut[threadIdx.x] = glut[threadIdx.x];
    __syncthreads();
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t p = idx[j];
    uint64_t* me64 = reinterpret_cast<uint64_t*>(s_rows + (size_t)threadIdx.x * pitch);
    const uint32_t w0 = s_offs[0] / 8u;                     // (the words this block of care offsets touches)
    for (uint32_t w = w0; w <= s_offs[cnt - 1] / 8u; w++) me64[w] = load_u64_unaligned(text + p + 8u * w);
    const uint8_t* me = reinterpret_cast<const uint8_t*>(me64);
    uint64_t k = 0;
    for (int c = 0; c < cnt; c++) {
        const uint32_t o = s_offs[c];
        const uint64_t code = p + o < n ? (uint64_t)(s_lut[me[o]] & 0x3ffu) : 0ull;
        k |= code << (64 - b * (c + 1));
    }
    key[j] = k;
}

// LCP of the mask arm of find_lcp (272-300): equal care characters while both sides are inside the text
__global__ void __launch_bounds__(256)
k_mask_lcp(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint32_t s,
           const uint32_t* __restrict__ offs, uint32_t weight, uint32_t* __restrict__ lcp)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= s) return;
    uint32_t c = 0;
    if (r > 0) {
        const uint64_t a = sa[r - 1], bpos = sa[r];
        while (c < weight) {
            const uint64_t qa = a + offs[c], qb = bpos + offs[c];
            if (qa >= n || qb >= n || text[qa] != text[qb]) break;
            c++;
        }
    }
    lcp[r] = c;
}

// The same with one visit of the text per rank: a workgroup stages the first `span` symbols (the reach of the mask) of
// its 256 suffixes and of the one ranked before them in LDS, eight bytes per load (the workspace copy of the text is
// padded), and every rank compares the care symbols with its neighbour's row there.
// (k_mask_lcp reads both suffixes of every pair, two sectors at random per rank, byte by byte: 298 ms of the 440 ms of
// the `hu-mask` build of the human-sized text, where weight 11 ties almost every neighbouring pair.)
static constexpr int MASK_SPAN_MAX = 64;     // symbols per row of the staged version
__global__ void __launch_bounds__(256)
k_mask_lcp_staged(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint32_t s,
                  const uint32_t* __restrict__ offs, uint32_t weight, uint32_t span, uint32_t* __restrict__ lcp)
{
    extern __shared__ __align__(16) uint8_t s_rows[];      // (256 + 1) rows of `pitch` bytes; row 0 = the rank before
    __shared__ uint32_t s_offs[MASK_SPAN_MAX];
    __shared__ uint8_t s_in[260];                          // care symbols of the row inside the text (the offsets ascend)
    const uint32_t words = (span + 7u) / 8u, pitch = words * 8u;
    if (threadIdx.x < weight) s_offs[threadIdx.x] = offs[threadIdx.x];
    __syncthreads();
    const uint32_t r0 = blockIdx.x * 256;
    for (uint32_t row = threadIdx.x; row < 257u; row += 256u) {
        const uint64_t r = (uint64_t)r0 + row;              // rank r0 - 1 + row
        if (r == 0 || r - 1 >= s) continue;
        const uint64_t p = sa[r - 1];
        uint64_t* me = reinterpret_cast<uint64_t*>(s_rows + (size_t)row * pitch);
        for (uint32_t w = 0; w < words; w++) me[w] = load_u64_unaligned(text + p + 8u * w);
        uint32_t in = 0;
        while (in < weight && p + s_offs[in] < n) in++;
        s_in[row] = (uint8_t)in;
    }
    __syncthreads();
    const uint32_t r = r0 + threadIdx.x;
    if (r >= s) return;
    uint32_t c = 0;
    if (r > 0) {
        const uint8_t* a = s_rows + (size_t)threadIdx.x * pitch;
        const uint8_t* b = a + pitch;
        const uint32_t lim = min((uint32_t)s_in[threadIdx.x], (uint32_t)s_in[threadIdx.x + 1]);
        while (c < lim && a[s_offs[c]] == b[s_offs[c]]) c++;
    }
    lcp[r] = c;
}

// The same count read off the SORTED KEYS (round 4): the key of a rank holds its first `cnt0` care symbols as codes from the top
// (k_mask_keys*: equal codes <=> equal bytes, 0 past the end of the text), so two neighbours part where their keys do -- a
// streaming pass over keys and positions instead of a random sector per rank (hu-mask on the 3.1 Gb stand-in: 96 -> 8 ms).
// A mask of more care symbols than a key holds continues in the text for the pairs whose keys are equal; a pair with a
// position within `span` of the end of the text counts symbol by symbol as before (past-the-end codes are equal, but do not count).
__global__ void __launch_bounds__(256)
k_mask_lcp_keys(const uint8_t* __restrict__ text, uint64_t n, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ sa,
                uint32_t s, const uint32_t* __restrict__ offs, uint32_t weight, uint32_t cnt0, int b, uint32_t span,
                uint32_t* __restrict__ lcp)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= s) return;
    uint32_t c = 0;
    if (r > 0) {
        const uint64_t a = sa[r - 1], p = sa[r];
        if ((a > p ? a : p) + span <= n) {
            const uint64_t x = keys[r - 1] ^ keys[r];
            c = x ? div_by_bits((uint32_t)__builtin_clzll(x), b) : cnt0;
            if (c > cnt0) c = cnt0;
            if (c == cnt0)
                while (c < weight && text[a + offs[c]] == text[p + offs[c]]) c++;
        } else {
            while (c < weight) {
                const uint64_t qa = a + offs[c], qb = p + offs[c];
                if (qa >= n || qb >= n || text[qa] != text[qb]) break;
                c++;
            }
        }
    }
    lcp[r] = c;
}

// Exact LCP of two suffixes of the device text: find_lcp(a, b, text_len, 0) of write()'s boundary fix
// (sufr_builder.rs:893-902), for the first record of a shard against the last record of the shard before it.
// One workgroup, 4096 characters per round.
// Under the order of the build (StitchOrder): a seed mask counts equal care symbols while both sides are inside the text
// (the mask arm of find_lcp, 272-300), --max-query-len caps the count (310-314).
struct StitchOrder {
    const uint32_t* offs;       // care offsets of the seed mask (device), or nullptr
    uint32_t weight;
    uint64_t cap;               // max_query_len, 0: none
};
__device__ __forceinline__ uint64_t masked_pair_lcp(const uint8_t* __restrict__ text, uint64_t n, uint64_t a, uint64_t b,
                                                    const StitchOrder& so)
{
    uint32_t c = 0;
    while (c < so.weight) {
        const uint64_t qa = a + so.offs[c], qb = b + so.offs[c];
        if (qa >= n || qb >= n || text[qa] != text[qb]) break;
        c++;
    }
    return c;
}

__global__ void __launch_bounds__(256)
k_lcp_pair(const uint8_t* __restrict__ text, uint64_t n, uint64_t a, uint64_t b, unsigned long long* __restrict__ out,
           StitchOrder so)
{
    __shared__ uint32_t s_first;
    if (so.offs) { if (threadIdx.x == 0) *out = masked_pair_lcp(text, n, a, b, so); return; }
    const uint64_t lim = n - (a > b ? a : b);          // characters both suffixes have
    for (uint64_t k = 0;; k += 4096) {
        if (threadIdx.x == 0) s_first = 0xffffffffu;
        __syncthreads();
        const uint64_t o = k + (uint64_t)threadIdx.x * 16;
        uint32_t m = 16;
        for (uint32_t i = 0; i < 16; i++) {
            const uint64_t q = o + i;
            if (q >= lim || text[a + q] != text[b + q]) { m = i; break; }
        }
        if (m < 16) atomicMin(&s_first, threadIdx.x * 16 + m);
        __syncthreads();
        const uint32_t f = s_first;
        __syncthreads();
        if (f != 0xffffffffu || (so.cap && k + 4096 >= so.cap)) {
            if (threadIdx.x == 0) { uint64_t v = f != 0xffffffffu ? k + f : so.cap; if (so.cap && v > so.cap) v = so.cap; *out = v; }
            return;
        }
    }
}

// The same boundary fix for a shard that stays on the device (N-GPU sort, one shard per rank): bounds[r] =
// {first suffix, last suffix, count} of every shard (the 24 bytes per rank of the all_gather, still in HBM);
// LCP[0] of shard `rank` := find_lcp(last suffix of the nearest non-empty shard before it, SA[0], text_len, 0).
// No host round trip: the pair is read from `bounds` here.
template <typename LT>                                     // uint32_t; uint64_t for the shards of a windowed build
__global__ void __launch_bounds__(256)
k_lcp_stitch(const uint8_t* __restrict__ text, uint64_t n, const unsigned long long* __restrict__ bounds, uint32_t rank,
             LT* __restrict__ lcp, StitchOrder so)
{
    __shared__ uint32_t s_first;
    if (bounds[(size_t)rank * 3 + 2] == 0ull) return;
    int prev = (int)rank - 1;
    while (prev >= 0 && bounds[(size_t)prev * 3 + 2] == 0ull) prev--;
    if (prev < 0) return;                                  // the globally first suffix: LCP 0, as the build left it
    const uint64_t a = bounds[(size_t)prev * 3 + 1], b = bounds[(size_t)rank * 3];
    if (so.offs) { if (threadIdx.x == 0) lcp[0] = (LT)masked_pair_lcp(text, n, a, b, so); return; }
    const uint64_t lim = n - (a > b ? a : b);          // characters both suffixes have
    for (uint64_t k = 0;; k += 4096) {
        if (threadIdx.x == 0) s_first = 0xffffffffu;
        __syncthreads();
        const uint64_t o = k + (uint64_t)threadIdx.x * 16;
        uint32_t m = 16;
        for (uint32_t i = 0; i < 16; i++) {
            const uint64_t q = o + i;
            if (q >= lim || text[a + q] != text[b + q]) { m = i; break; }
        }
        if (m < 16) atomicMin(&s_first, threadIdx.x * 16 + m);
        __syncthreads();
        const uint32_t f = s_first;
        __syncthreads();
        if (f != 0xffffffffu || (so.cap && k + 4096 >= so.cap)) {
            if (threadIdx.x == 0) { uint64_t v = f != 0xffffffffu ? k + f : so.cap; if (so.cap && v > so.cap) v = so.cap; lcp[0] = (LT)v; }
            return;
        }
    }
}

// widen u32 results for the u64-index ABI (texts below 2^32-1 only)
__global__ void __launch_bounds__(256)
k_widen(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = in[i];
}

}  // namespace sufr

#include "sufr_msd.inc"
#include "sufr_part.inc"
#include "sufr_dbl.inc"
#include "sufr_runs.inc"
#include "sufr_launch.inc"
#include "sufr_wide.inc"
#include "sufr_exc.inc"
#include "sufr_capi.inc"
#include "../../include/sufr_query.h"
#include "sufr_search.inc"
