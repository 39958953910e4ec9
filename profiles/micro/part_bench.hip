// part_bench.hip -- stand-alone probe of radix-partition kernel designs (level 1: bit-packed text -> buckets of
// (key, index) records on the first 5 characters).  Not part of the library; it exists to compare store
// patterns on the hardware before one of them goes into sufr_msd.inc.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o part_bench part_bench.hip && ./part_bench [n_positions] [groups]
// Synthetic text: 3-bit codes {$=1 %=2 A=3 C=4 G=5 N=6 T=7}, uniform ACGT with ~50 % of the positions inside N runs
// (mean 350), which is what --ignore-softmask makes of a soft-masked genome.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int WAVE = 64;
static constexpr int B = 3, K = 21, DB_ = 15;          // bits per code, codes per key, digit bits
static constexpr uint32_t ELIG = (1u << 1) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 7);
static constexpr uint32_t MAXB = 4096;

__host__ __device__ inline uint64_t mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint32_t code_at(uint64_t p, uint64_t n)
{
    if (p >= n) return 0;
    if (p == n - 1) return 1;
    const uint64_t blk = p / 700, off = p % 700;
    const uint64_t L = mix(blk * 2 + 1) % 701;
    if (off >= L) return 6;
    const uint32_t r = (uint32_t)(mix(p * 2) >> 20) & 3u;
    return r == 0 ? 3u : (r == 1 ? 4u : (r == 2 ? 5u : 7u));
}
__global__ void k_gen(uint8_t* packed, uint64_t n, uint64_t nthreads)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    uint64_t V = 0;
    for (int e = 0; e < 16; e++) V = (V << 3) | code_at(t * 16 + e, n);
    uint8_t* q = packed + t * 6;
    for (int k = 0; k < 6; k++) q[k] = (uint8_t)(V >> (40 - 8 * k));
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t ld64u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

template <int E>
__device__ __forceinline__ void build_keys(const uint8_t* __restrict__ packed, uint64_t pos0, uint64_t (&key)[E], uint32_t& elig)
{
    constexpr uint64_t keep = ~0ull << (64 - K * B);
    const uint8_t* pp = packed + ((pos0 * B) >> 3);
    const uint64_t hi = __builtin_bswap64(ld64u(pp));
    const uint64_t lo = __builtin_bswap64(ld64u(pp + 8));
    elig = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int s = e * B;
        const uint64_t v = s ? ((hi << s) | (lo >> (64 - s))) : hi;
        key[e] = v & keep;
        elig |= ((ELIG >> (uint32_t)(v >> (64 - B))) & 1u) << e;
    }
}

struct DigitMap { const uint64_t* pres; const uint16_t* rowbase; uint32_t rows, nbins; };
struct LdsMap { const uint64_t* pres; const uint16_t* rowbase; };
__host__ __device__ inline size_t map_bytes(uint32_t rows) { return (size_t)rows * 8 + (((size_t)rows * 2 + 15) & ~(size_t)15); }
template <int NT>
__device__ __forceinline__ LdsMap load_map(const DigitMap& dm, uint8_t* area)
{
    uint64_t* sp = reinterpret_cast<uint64_t*>(area);
    uint16_t* sr = reinterpret_cast<uint16_t*>(area + (size_t)dm.rows * 8);
    for (uint32_t i = threadIdx.x; i < dm.rows; i += NT) { sp[i] = dm.pres[i]; sr[i] = dm.rowbase[i]; }
    LdsMap m; m.pres = sp; m.rowbase = sr; return m;
}
__device__ __forceinline__ uint32_t dense_digit(uint32_t raw, const LdsMap& m)
{
    const uint32_t hi = raw >> 6, lo = raw & 63u;
    return (uint32_t)m.rowbase[hi] + (uint32_t)__popcll(m.pres[hi] & ((1ull << lo) - 1ull));
}

__global__ void __launch_bounds__(256) k_presence(const uint8_t* packed, uint64_t n, uint32_t* flags)
{
    const uint64_t pos0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (pos0 >= n) return;
    uint64_t key[16]; uint32_t el;
    build_keys<16>(packed, pos0, key, el);
#pragma unroll
    for (int e = 0; e < 16; e++) if (pos0 + e < n) flags[(uint32_t)(key[e] >> (64 - DB_))] = 1u;
}

__global__ void __launch_bounds__(256) k_hist(const uint8_t* packed, uint64_t n, DigitMap dm, uint64_t chunk, uint32_t G,
                                              uint32_t* grouptab, unsigned long long* sums)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const LdsMap map = load_map<256>(dm, smem);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + map_bytes(dm.rows));
    for (uint32_t i = threadIdx.x; i < dm.nbins; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(c0 + chunk, n);
    unsigned long long si = 0, sq = 0;
    for (uint64_t t0 = c0; t0 < c1; t0 += 4096) {
        uint64_t key[16]; uint32_t el;
        const uint64_t pos0 = t0 + (uint64_t)threadIdx.x * 16;
        build_keys<16>(packed, pos0, key, el);
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (el & (1u << e)) {
                atomicAdd(&s_cnt[dense_digit((uint32_t)(key[e] >> (64 - DB_)), map)], 1u);
                si += pos0 + e; sq += (pos0 + e) * (pos0 + e) + key[e];
            }
    }
    __syncthreads();
    uint32_t* row = grouptab + (size_t)(blockIdx.x % G) * dm.nbins;
    for (uint32_t i = threadIdx.x; i < dm.nbins; i += 256) if (s_cnt[i]) atomicAdd(&row[i], s_cnt[i]);
    atomicAdd(&sums[0], si); atomicAdd(&sums[1], sq);
}

struct Rec { uint32_t klo, khi, idx; };

// ---------------------------------------------------------------------------------------------------------------
// V0: the kernel as it is in sufr_msd.inc today -- records staged in LDS in digit order, runs copied out (two arrays)
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
struct Claim { static constexpr int PB = (int)MAXB / NT; uint32_t base[PB]; };
template <int NT>
__device__ __forceinline__ void claim_begin(uint32_t* s_cnt, uint32_t NB, uint32_t* __restrict__ cur, uint32_t* s_wsum,
                                            uint32_t* s_total, Claim<NT>& cl)
{
    constexpr int PB = Claim<NT>::PB;
    const uint32_t d0 = threadIdx.x * PB;
    uint32_t c[PB]; uint32_t local = 0;
#pragma unroll
    for (int k = 0; k < PB; k++) { c[k] = (d0 + k < NB) ? s_cnt[d0 + k] : 0u; local += c[k]; }
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = __shfl_up(incl, o, WAVE); if ((int)lane_id() >= o) incl += t; }
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += s_wsum[w];
    if (threadIdx.x == NT - 1) *s_total = wbase + incl;
    uint32_t run = wbase + incl - local;
#pragma unroll
    for (int k = 0; k < PB; k++) { if (d0 + k < NB) s_cnt[d0 + k] = run; run += c[k]; }
    __syncthreads();
    const uint32_t total = *s_total;
#pragma unroll
    for (int k = 0; k < PB; k++) {
        const uint32_t d = (uint32_t)k * NT + threadIdx.x;
        cl.base[k] = 0;
        if (d < NB) {
            const uint32_t cnt = (d + 1 < NB ? s_cnt[d + 1] : total) - s_cnt[d];
            if (cnt) cl.base[k] = atomicAdd(&cur[d], cnt);
        }
    }
}
template <int NT>
__device__ __forceinline__ void claim_end(uint32_t* s_cnt, uint32_t NB, const Claim<NT>& cl)
{
    constexpr int PB = Claim<NT>::PB;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PB; k++) { const uint32_t d = (uint32_t)k * NT + threadIdx.x; if (d < NB) s_cnt[d] = cl.base[k] - s_cnt[d]; }
    __syncthreads();
}

template <bool AOS, int NT, int E, int CAP, int MINW, int ABL>
__global__ void __launch_bounds__(NT, MINW)
k_staged(uint64_t n, const uint8_t* __restrict__ packed, DigitMap dm, uint64_t chunk, uint32_t ngroups,
         uint32_t* __restrict__ gcur, uint64_t* __restrict__ out_key, uint32_t* __restrict__ out_idx, Rec* __restrict__ out_rec)
{
    constexpr int TILEB = NT * E;
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t NB = dm.nbins, NBa = (NB + 3u) & ~3u;
    uint64_t* s_key = reinterpret_cast<uint64_t*>(smem);
    const LdsMap map = load_map<NT>(dm, smem + (size_t)CAP * 8);
    uint32_t* s_pv = reinterpret_cast<uint32_t*>(smem + (size_t)CAP * 8 + map_bytes(dm.rows));
    uint32_t* s_cnt = s_pv + CAP;
    uint32_t* s_misc = s_cnt + NBa;
    uint32_t* cur = gcur + (size_t)(blockIdx.x % ngroups) * NB;
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(c0 + chunk, n);
    for (uint64_t tile0 = c0; tile0 < c1; tile0 += TILEB) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < NB; i += NT) s_cnt[i] = 0;
        uint64_t key[E]; uint32_t keep;
        build_keys<E>(packed, tile0 + (uint64_t)threadIdx.x * E, key, keep);
        uint32_t dig[E];
#pragma unroll
        for (int e = 0; e < E; e++) dig[e] = (keep & (1u << e)) ? dense_digit((uint32_t)(key[e] >> (64 - DB_)), map) : 0u;
        {
            uint32_t c = (uint32_t)__popc(keep);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, WAVE);
            if (lane_id() == 0) s_misc[8 + (threadIdx.x >> 6)] = c;
        }
        __syncthreads();
        uint32_t kept = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) kept += s_misc[8 + w];
        const int halves = kept > (uint32_t)CAP ? 2 : 1;
        for (int h = 0; h < halves; h++) {
            const uint32_t mask = halves == 1 ? keep : (keep & (h ? (0xffffffffu << (E / 2)) : ((1u << (E / 2)) - 1u)));
            if (h) { __syncthreads(); for (uint32_t i = threadIdx.x; i < NB; i += NT) s_cnt[i] = 0; __syncthreads(); }
#pragma unroll
            for (int e = 0; e < E; e++)
                if (mask & (1u << e)) dig[e] = (dig[e] & 0xfffu) | (atomicAdd(&s_cnt[dig[e] & 0xfffu], 1u) << 12);
            __syncthreads();
            Claim<NT> cl;
            claim_begin<NT>(s_cnt, NB, cur, s_misc, s_misc + 24, cl);
            const uint32_t total = s_misc[24];
#pragma unroll
            for (int e = 0; e < E; e++)
                if (mask & (1u << e)) {
                    const uint32_t d = dig[e] & 0xfffu;
                    const uint32_t pos = s_cnt[d] + (dig[e] >> 12);
                    s_key[pos] = key[e];
                    s_pv[pos] = (d << 14) | (uint32_t)(threadIdx.x * E + e);
                }
            claim_end<NT>(s_cnt, NB, cl);
            for (uint32_t j = threadIdx.x; j < total; j += NT) {
                const uint32_t v = s_pv[j];
                uint32_t o = j + s_cnt[v >> 14];
                const uint64_t k = s_key[j];
                const uint32_t ix = (uint32_t)(tile0 + (v & 0x3fffu));
                if (ABL == 1) { if (k == 0x123456789ull) out_idx[o] = ix; continue; }     // no global stores
                if (ABL == 2) o = (uint32_t)((tile0 >> 1) + j);                            // dense stores, same bytes
                if (AOS) { Rec r; r.klo = (uint32_t)k; r.khi = (uint32_t)(k >> 32); r.idx = ix; out_rec[o] = r; }
                else { out_key[o] = k; out_idx[o] = ix; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// V3: big tiles.  The tile's packed text lies in LDS (as big-endian dwords), the records of a tile are staged as
// 2-byte positions in digit order, and the copy-out cuts every record's key out of the LDS text again.  With 2 bytes
// of staging per record a tile holds 32768 positions: four times the records per digit run, a quarter of the
// per-tile bin work (scan, claims, barriers) per record.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int RNDS, int CAP, int MINW, int ABL = 0, int STAG = 0>
__global__ void __launch_bounds__(NT, MINW)
k_bigtile(uint64_t n, const uint8_t* __restrict__ packed, DigitMap dm, uint64_t chunk, uint32_t ngroups,
          uint32_t* __restrict__ gcur, uint64_t* __restrict__ out_key, uint32_t* __restrict__ out_idx, Rec* __restrict__ out_rec)
{
    constexpr int P = NT * 32 * RNDS;                   // positions per tile
    constexpr int TW = P * B / 32;                      // dwords of packed text per tile
    constexpr int PB = (int)MAXB / NT;
    static_assert(P <= 65536 && CAP >= P / 2 && RNDS % 2 == 0, "tile geometry");
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t NB = dm.nbins, NBa = (NB + 3u) & ~3u;
    uint32_t* s_text = reinterpret_cast<uint32_t*>(smem);                               // TW + 4
    const LdsMap map = load_map<NT>(dm, smem + (size_t)(TW + 4) * 4);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + (size_t)(TW + 4) * 4 + map_bytes(dm.rows));   // counts -> slot delta
    uint32_t* s_start = s_cnt + NBa;                                                    // exclusive prefix -> running slot
    uint32_t* s_misc = s_start + NBa;                                                   // 32
    uint16_t* s_pos = reinterpret_cast<uint16_t*>(s_misc + 32);                         // CAP
    uint32_t* cur = gcur + (size_t)(blockIdx.x % ngroups) * NB;
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(c0 + chunk, n);
    const int wv = threadIdx.x >> 6;
    constexpr int NPRE = (TW / 4 + 1 + NT - 1) / NT;
    uint4 nxt[NPRE];
    auto fetch = [&](uint64_t t0) {
        const uint4* src = reinterpret_cast<const uint4*>(packed + ((t0 * B) >> 3));
#pragma unroll
        for (int u = 0; u < NPRE; u++) { const int v = threadIdx.x + u * NT; nxt[u] = v < TW / 4 + 1 ? src[v] : make_uint4(0, 0, 0, 0); }
    };
    auto swap_pin = [&]() {
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            nxt[u].x = __builtin_bswap32(nxt[u].x); nxt[u].y = __builtin_bswap32(nxt[u].y);
            nxt[u].z = __builtin_bswap32(nxt[u].z); nxt[u].w = __builtin_bswap32(nxt[u].w);
            asm volatile("" : "+v"(nxt[u].x), "+v"(nxt[u].y), "+v"(nxt[u].z), "+v"(nxt[u].w));
        }
    };
    fetch(c0);
    swap_pin();
    // STAG: every other resident workgroup starts with a half tile, so that the store phases of the workgroups
    // that share a CU (and of the chip as a whole) do not coincide
    uint32_t rounds = (STAG && ((blockIdx.x >> STAG) & 1u)) ? RNDS / 2 : RNDS;
    for (uint64_t tile0 = c0; tile0 < c1; tile0 += (uint64_t)rounds * NT * 32, rounds = RNDS) {
        __syncthreads();                                 // the previous tile is out of the LDS
        {
            uint4* dst = reinterpret_cast<uint4*>(s_text);
#pragma unroll
            for (int u = 0; u < NPRE; u++) { const int v = threadIdx.x + u * NT; if (v < TW / 4 + 1) dst[v] = nxt[u]; }
        }
        __syncthreads();
        // suffix starts of this thread (eligibility is the first character's)
        uint32_t elig[RNDS];
        uint32_t mykept = 0;
#pragma unroll
        for (int r = 0; r < RNDS; r++) {
            const uint32_t* tw = s_text + (size_t)(r * NT + threadIdx.x) * 3;
            const uint32_t w0 = tw[0], w1 = tw[1], w2 = tw[2];
            // code c is eligible iff bit c of ELIG: count over the 32 codes of the three words
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 32; e++) {
                const int bit = 3 * e, i = bit >> 5, sh = bit & 31;
                const uint32_t wa = i == 0 ? w0 : (i == 1 ? w1 : w2), wb = i == 0 ? w1 : (i == 1 ? w2 : 0u);
                const uint32_t x = sh ? __builtin_amdgcn_alignbit(wa, wb, 32 - sh) : wa;
                m |= ((ELIG >> (x >> 29)) & 1u) << e;
            }
            if ((uint32_t)r >= rounds || tile0 + (uint64_t)r * NT * 32 >= c1) m = 0;
            elig[r] = m;
            mykept += (uint32_t)__popc(m);
        }
        {
            uint32_t c = mykept;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, WAVE);
            if (lane_id() == 0) s_misc[8 + wv] = c;
        }
        for (uint32_t i = threadIdx.x; i < NB; i += NT) s_cnt[i] = 0;
        __syncthreads();
        uint32_t kept = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) kept += s_misc[8 + w];
        const int halves = kept > (uint32_t)CAP ? 2 : 1;
        for (int h = 0; h < halves; h++) {
            const int r0 = halves == 1 ? 0 : h * (RNDS / 2), r1 = halves == 1 ? RNDS : (h + 1) * (RNDS / 2);
            if (h) { __syncthreads(); for (uint32_t i = threadIdx.x; i < NB; i += NT) s_cnt[i] = 0; __syncthreads(); }
            if (ABL != 4)
#pragma unroll
            for (int r = 0; r < RNDS; r++)
                if (r >= r0 && r < r1 && elig[r]) {
                    const uint32_t* tw = s_text + (size_t)(r * NT + threadIdx.x) * 3;
                    uint64_t hi = ((uint64_t)tw[0] << 32) | tw[1], lo = ((uint64_t)tw[2] << 32) | tw[3];
                    uint32_t m = elig[r];
#pragma unroll 8
                    for (int e = 0; e < 32; e++) {
                        if (m & 1u) atomicAdd(&s_cnt[dense_digit((uint32_t)(hi >> 49), map)], 1u);
                        m >>= 1; hi = (hi << 3) | (lo >> 61); lo <<= 3;
                    }
                }
            __syncthreads();
            // exclusive scan of the counts (PB consecutive bins per thread) -> s_start; claims on bins k * NT + thread
            uint32_t base[PB];
            {
                const uint32_t d0 = threadIdx.x * PB;
                uint32_t c[PB]; uint32_t local = 0;
#pragma unroll
                for (int k = 0; k < PB; k++) { c[k] = (d0 + k < NB) ? s_cnt[d0 + k] : 0u; local += c[k]; }
                uint32_t incl = local;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = __shfl_up(incl, o, WAVE); if ((int)lane_id() >= o) incl += t; }
                if (lane_id() == 63) s_misc[wv] = incl;
                __syncthreads();
                uint32_t wbase = 0;
                for (int w = 0; w < wv; w++) wbase += s_misc[w];
                if (threadIdx.x == NT - 1) s_misc[24] = wbase + incl;
                uint32_t run = wbase + incl - local;
#pragma unroll
                for (int k = 0; k < PB; k++) { if (d0 + k < NB) s_start[d0 + k] = run; run += c[k]; }
#pragma unroll
                for (int k = 0; k < PB; k++) {
                    const uint32_t d = (uint32_t)k * NT + threadIdx.x;
                    base[k] = 0;
                    if (d < NB) { const uint32_t cnt = s_cnt[d]; if (cnt) base[k] = atomicAdd(&cur[d], cnt); }
                }
                __syncthreads();
            }
            const uint32_t total = s_misc[24];
            if (h == halves - 1 && tile0 + (uint64_t)rounds * NT * 32 < c1) fetch(tile0 + (uint64_t)rounds * NT * 32);
            if (ABL == 3 || ABL == 4) continue;
            // every record takes the next slot of its digit
#pragma unroll
            for (int r = 0; r < RNDS; r++)
                if (r >= r0 && r < r1 && elig[r]) {
                    const uint32_t p0 = (uint32_t)(r * NT + threadIdx.x) * 32u;
                    const uint32_t* tw = s_text + (size_t)(r * NT + threadIdx.x) * 3;
                    uint64_t hi = ((uint64_t)tw[0] << 32) | tw[1], lo = ((uint64_t)tw[2] << 32) | tw[3];
                    uint32_t m = elig[r];
#pragma unroll 8
                    for (int e = 0; e < 32; e++) {
                        if (m & 1u) s_pos[atomicAdd(&s_start[dense_digit((uint32_t)(hi >> 49), map)], 1u)] = (uint16_t)(p0 + e);
                        m >>= 1; hi = (hi << 3) | (lo >> 61); lo <<= 3;
                    }
                }
            __syncthreads();
            // slot delta of every digit: claimed base - first slot (s_start[d] now holds the first slot of d + 1)
#pragma unroll
            for (int k = 0; k < PB; k++) {
                const uint32_t d = (uint32_t)k * NT + threadIdx.x;
                if (d < NB) s_cnt[d] = base[k] - (d ? s_start[d - 1] : 0u);
            }
            if (h == halves - 1 && tile0 + (uint64_t)rounds * NT * 32 < c1) swap_pin();
            __syncthreads();
            if (ABL == 2) continue;
            for (uint32_t j0 = threadIdx.x; j0 < total; j0 += 4 * NT) {
                uint32_t pos[4], w0[4], w1[4], w2[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t j = j0 + u * NT; pos[u] = j < total ? s_pos[j] : 0u; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t i = (pos[u] * 3u) >> 5; w0[u] = s_text[i]; w1[u] = s_text[i + 1]; w2[u] = s_text[i + 2]; }
                uint32_t khi[4], klo[4], dd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t sh = (pos[u] * 3u) & 31u;
                    khi[u] = (uint32_t)((((uint64_t)w0[u] << 32) | w1[u]) >> (32u - sh));
                    klo[u] = (uint32_t)((((uint64_t)w1[u] << 32) | w2[u]) >> (32u - sh)) & ~1u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) dd[u] = dense_digit(khi[u] >> 17, map);
#pragma unroll
                for (int u = 0; u < 4; u++) dd[u] = s_cnt[dd[u]];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + u * NT;
                    if (j < total) {
                        Rec rr; rr.klo = klo[u]; rr.khi = khi[u]; rr.idx = (uint32_t)(tile0 + pos[u]);
                        if (ABL == 1) { if (klo[u] == 0x12345u && dd[u] == 77u) out_rec[j + dd[u]] = rr; }
                        else if (ABL == 5) out_idx[j + dd[u]] = rr.idx;                                     // 4 bytes per record
                        else if (ABL == 6) reinterpret_cast<uint4*>(out_key)[j + dd[u]] = make_uint4(rr.klo, rr.khi, rr.idx, 0u);   // 16 bytes
                        else if (ABL == 7) out_rec[(tile0 >> 1) + j] = rr;                                   // dense
                        else if (ABL == 8) { out_key[j + dd[u]] = ((uint64_t)rr.khi << 32) | rr.klo; out_idx[j + dd[u]] = rr.idx; }   // two arrays
                        else out_rec[j + dd[u]] = rr;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// V4: big tiles, rank kept in registers: one returning LDS atomic per record (pass A), no second atomic pass
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int RNDS, int CAP, int MINW, int SB>
__global__ void __launch_bounds__(NT, MINW)
k_bigtile2(uint64_t n, const uint8_t* __restrict__ packed, DigitMap dm, uint64_t chunk, uint32_t ngroups,
           uint32_t* __restrict__ gcur, uint64_t* __restrict__ out_key, uint32_t* __restrict__ out_idx, Rec* __restrict__ out_rec)
{
    constexpr int P = NT * 32 * RNDS;
    constexpr int TW = P * B / 32;
    constexpr int PB = (int)MAXB / NT;
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t NB = dm.nbins, NBa = (NB + 3u) & ~3u;
    uint32_t* s_text = reinterpret_cast<uint32_t*>(smem);
    const LdsMap map = load_map<NT>(dm, smem + (size_t)(TW + 4) * 4);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + (size_t)(TW + 4) * 4 + map_bytes(dm.rows));
    uint32_t* s_start = s_cnt + NBa;
    uint32_t* s_misc = s_start + NBa;
    uint16_t* s_pos = reinterpret_cast<uint16_t*>(s_misc + 32);
    uint32_t* cur = gcur + (size_t)(blockIdx.x % ngroups) * NB;
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(c0 + chunk, n);
    const int wv = threadIdx.x >> 6;
    constexpr int NPRE = (TW / 4 + 1 + NT - 1) / NT;
    uint4 nxt[NPRE];
    auto fetch = [&](uint64_t t0) {
        const uint4* src = reinterpret_cast<const uint4*>(packed + ((t0 * B) >> 3));
#pragma unroll
        for (int u = 0; u < NPRE; u++) { const int v = threadIdx.x + u * NT; nxt[u] = v < TW / 4 + 1 ? src[v] : make_uint4(0, 0, 0, 0); }
    };
    auto swap_pin = [&]() {
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            nxt[u].x = __builtin_bswap32(nxt[u].x); nxt[u].y = __builtin_bswap32(nxt[u].y);
            nxt[u].z = __builtin_bswap32(nxt[u].z); nxt[u].w = __builtin_bswap32(nxt[u].w);
            asm volatile("" : "+v"(nxt[u].x), "+v"(nxt[u].y), "+v"(nxt[u].z), "+v"(nxt[u].w));
        }
    };
    fetch(c0);
    swap_pin();
    for (uint64_t tile0 = c0; tile0 < c1; tile0 += P) {
        __syncthreads();
        {
            uint4* dst = reinterpret_cast<uint4*>(s_text);
#pragma unroll
            for (int u = 0; u < NPRE; u++) { const int v = threadIdx.x + u * NT; if (v < TW / 4 + 1) dst[v] = nxt[u]; }
        }
        for (uint32_t i = threadIdx.x; i < NB; i += NT) s_cnt[i] = 0;
        __syncthreads();
        // pass A: digit and rank (inside the tile) of every suffix start of this thread
        uint32_t dr[RNDS * 32];
        uint32_t elig[RNDS];
#pragma unroll
        for (int r = 0; r < RNDS; r++) {
            const uint32_t* tw = s_text + (size_t)(r * NT + threadIdx.x) * 3;
            uint64_t hi = ((uint64_t)tw[0] << 32) | tw[1], lo = ((uint64_t)tw[2] << 32) | tw[3];
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 32; e++) {
                const uint32_t top = (uint32_t)(hi >> 49);
                const bool el = (ELIG >> (top >> 12)) & 1u;
                dr[r * 32 + e] = 0;
                if (el) {
                    const uint32_t d = dense_digit(top, map);
                    dr[r * 32 + e] = d | (atomicAdd(&s_cnt[d], 1u) << 12);
                    m |= 1u << e;
                }
                hi = (hi << 3) | (lo >> 61); lo <<= 3;
                if (SB && (e % SB) == SB - 1) __builtin_amdgcn_sched_barrier(0);
            }
            elig[r] = m;
        }
        __syncthreads();
        uint32_t base[PB];
        {
            const uint32_t d0 = threadIdx.x * PB;
            uint32_t c[PB]; uint32_t local = 0;
#pragma unroll
            for (int k = 0; k < PB; k++) { c[k] = (d0 + k < NB) ? s_cnt[d0 + k] : 0u; local += c[k]; }
            uint32_t incl = local;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = __shfl_up(incl, o, WAVE); if ((int)lane_id() >= o) incl += t; }
            if (lane_id() == 63) s_misc[wv] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (int w = 0; w < wv; w++) wbase += s_misc[w];
            if (threadIdx.x == NT - 1) s_misc[24] = wbase + incl;
            uint32_t run = wbase + incl - local;
#pragma unroll
            for (int k = 0; k < PB; k++) { if (d0 + k < NB) s_start[d0 + k] = run; run += c[k]; }
#pragma unroll
            for (int k = 0; k < PB; k++) {
                const uint32_t d = (uint32_t)k * NT + threadIdx.x;
                base[k] = 0;
                if (d < NB) { const uint32_t cnt = s_cnt[d]; if (cnt) base[k] = atomicAdd(&cur[d], cnt); }
            }
            __syncthreads();
        }
        const uint32_t total = s_misc[24];          // (the probe assumes total <= CAP)
        if (tile0 + P < c1) fetch(tile0 + P);
#pragma unroll
        for (int r = 0; r < RNDS; r++) {
            const uint32_t p0 = (uint32_t)(r * NT + threadIdx.x) * 32u;
#pragma unroll
            for (int e = 0; e < 32; e++)
                if (elig[r] & (1u << e)) s_pos[s_start[dr[r * 32 + e] & 0xfffu] + (dr[r * 32 + e] >> 12)] = (uint16_t)(p0 + e);
        }
#pragma unroll
        for (int k = 0; k < PB; k++) {
            const uint32_t d = (uint32_t)k * NT + threadIdx.x;
            if (d < NB) s_cnt[d] = base[k] - s_start[d];
        }
        if (tile0 + P < c1) swap_pin();
        __syncthreads();
        for (uint32_t j0 = threadIdx.x; j0 < total; j0 += 4 * NT) {
            uint32_t pos[4], w0[4], w1[4], w2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t j = j0 + u * NT; pos[u] = j < total ? s_pos[j] : 0u; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t i = (pos[u] * 3u) >> 5; w0[u] = s_text[i]; w1[u] = s_text[i + 1]; w2[u] = s_text[i + 2]; }
            uint32_t khi[4], klo[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t sh = (pos[u] * 3u) & 31u;
                khi[u] = (uint32_t)((((uint64_t)w0[u] << 32) | w1[u]) >> (32u - sh));
                klo[u] = (uint32_t)((((uint64_t)w1[u] << 32) | w2[u]) >> (32u - sh)) & ~1u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) dd[u] = dense_digit(khi[u] >> 17, map);
#pragma unroll
            for (int u = 0; u < 4; u++) dd[u] = s_cnt[dd[u]];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = j0 + u * NT;
                if (j < total) { Rec rr; rr.klo = klo[u]; rr.khi = khi[u]; rr.idx = (uint32_t)(tile0 + pos[u]); out_rec[j + dd[u]] = rr; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// verification + reference streams
// ---------------------------------------------------------------------------------------------------------------
__device__ inline uint64_t key_at(const uint8_t* packed, uint64_t p)
{
    uint64_t k = 0;
    const uint64_t bit = p * 3;
    const uint8_t* q = packed + (bit >> 3);
    // 63 bits from bit offset (bit & 7): read 9 bytes
    unsigned __int128 w = 0;
    for (int i = 0; i < 9; i++) w = (w << 8) | q[i];
    w <<= (bit & 7);                     // drop leading bits (now top of 72-bit window)
    k = (uint64_t)(w >> 8);              // top 64 bits of the 72-bit window
    return k & (~0ull << 1);
}
template <bool AOS>
__global__ void __launch_bounds__(256)
k_verify(const uint8_t* packed, DigitMap dm, const uint32_t* leafbase, uint64_t s, const uint64_t* out_key,
         const uint32_t* out_idx, const Rec* out_rec, unsigned long long* sums, unsigned long long* bad)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const LdsMap map = load_map<256>(dm, smem);
    __syncthreads();
    unsigned long long si = 0, sq = 0, nb = 0;
    for (uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x; o < s; o += (uint64_t)gridDim.x * 256) {
        uint64_t k; uint32_t ix;
        if (AOS) { const Rec r = out_rec[o]; k = ((uint64_t)r.khi << 32) | r.klo; ix = r.idx; }
        else { k = out_key[o]; ix = out_idx[o]; }
        const uint32_t d = dense_digit((uint32_t)(k >> (64 - DB_)), map);
        if (!(d < dm.nbins && leafbase[d] <= o && o < leafbase[d + 1])) { nb++; atomicAdd(&sums[5], 1ull); }
        if (key_at(packed, ix) != k) { nb++; atomicAdd(&sums[6], 1ull); }
        si += ix; sq += (unsigned long long)ix * ix + k;
    }
    atomicAdd(&sums[0], si); atomicAdd(&sums[1], sq); if (nb) atomicAdd(bad, nb);
}
__global__ void __launch_bounds__(256) k_fill(uint4* p, uint64_t nvec)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * 256)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) k_copy(const uint4* a, uint4* p, uint64_t nvec)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * 256) p[i] = a[i];
}

int main(int argc, char** argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 3100000001ull;
    const uint32_t G = argc > 2 ? (uint32_t)atoi(argv[2]) : 8u;
    const char* only = argc > 3 ? argv[3] : "";
    const uint64_t nthreads = (n + 15) / 16;
    uint8_t* packed; const size_t pbytes = nthreads * 6 + 8192;
    CK(hipMalloc(&packed, pbytes)); CK(hipMemset(packed, 0, pbytes));
    hipLaunchKernelGGL(k_gen, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, 0, packed, n, nthreads);
    CK(hipDeviceSynchronize());
    // presence -> digit map
    uint32_t* flags; CK(hipMalloc(&flags, 32768 * 4)); CK(hipMemset(flags, 0, 32768 * 4));
    hipLaunchKernelGGL(k_presence, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, 0, packed, n, flags);
    std::vector<uint32_t> hf(32768); CK(hipMemcpy(hf.data(), flags, 32768 * 4, hipMemcpyDeviceToHost));
    hf[0] = 1;
    const uint32_t rows = 512;
    std::vector<uint8_t> hm(map_bytes(rows), 0);
    uint64_t* pres = (uint64_t*)hm.data(); uint16_t* rb = (uint16_t*)(hm.data() + rows * 8);
    uint32_t NB = 0;
    for (uint32_t r = 0; r < rows; r++) { rb[r] = (uint16_t)NB; for (uint32_t i = 0; i < 64; i++) if (hf[r * 64 + i]) { pres[r] |= 1ull << i; NB++; } }
    uint8_t* dmap; CK(hipMalloc(&dmap, hm.size())); CK(hipMemcpy(dmap, hm.data(), hm.size(), hipMemcpyHostToDevice));
    DigitMap dm; dm.pres = (const uint64_t*)dmap; dm.rowbase = (const uint16_t*)(dmap + rows * 8); dm.rows = rows; dm.nbins = NB;
    printf("n=%llu groups=%u dense digits=%u\n", (unsigned long long)n, G, NB);
    // chunking like the library: 2048 workgroups, chunk a multiple of 16384 positions
    const uint64_t tiles = (n + 4095) / 4096;
    uint64_t nwg = 2048; uint64_t per = (tiles + nwg - 1) / nwg; per = (per + 15) & ~15ull;
    const uint64_t chunk = per * 4096; nwg = (n + chunk - 1) / chunk;
    uint32_t *grouptab, *gcur, *gcur0, *leafbase; unsigned long long* sums;
    CK(hipMalloc(&grouptab, (size_t)G * NB * 4)); CK(hipMemset(grouptab, 0, (size_t)G * NB * 4));
    CK(hipMalloc(&gcur, (size_t)G * NB * 4)); CK(hipMalloc(&gcur0, (size_t)G * NB * 4)); CK(hipMalloc(&leafbase, (NB + 1) * 4));
    CK(hipMalloc(&sums, 64)); CK(hipMemset(sums, 0, 64));
    hipLaunchKernelGGL(k_hist, dim3((uint32_t)nwg), dim3(256), map_bytes(rows) + NB * 4, 0, packed, n, dm, chunk, G, grouptab, sums);
    std::vector<uint32_t> gt((size_t)G * NB); CK(hipMemcpy(gt.data(), grouptab, gt.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long want[2]; CK(hipMemcpy(want, sums, 16, hipMemcpyDeviceToHost));
    std::vector<uint32_t> lb(NB + 1), gc((size_t)G * NB);
    uint64_t run = 0;
    for (uint32_t d = 0; d < NB; d++) { lb[d] = (uint32_t)run; for (uint32_t g = 0; g < G; g++) { gc[(size_t)g * NB + d] = (uint32_t)run; run += gt[(size_t)g * NB + d]; } }
    lb[NB] = (uint32_t)run;
    const uint64_t s = run;
    CK(hipMemcpy(leafbase, lb.data(), (NB + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gcur0, gc.data(), gc.size() * 4, hipMemcpyHostToDevice));
    printf("s=%llu (%.1f %% of n), workgroups=%llu chunk=%llu\n", (unsigned long long)s, 100.0 * s / n, (unsigned long long)nwg, (unsigned long long)chunk);
    uint64_t* okey; uint32_t* oidx; Rec* orec;
    CK(hipMalloc(&okey, s * 16 + 256)); CK(hipMalloc(&oidx, s * 4 + 256)); CK(hipMalloc(&orec, s * 12 + 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg = (double)n + 4.0 * s, phys = 3.0 * n / 8 + 12.0 * s;

    auto report = [&](const char* name, bool aos, float best, float avg) {
        // verify the last run
        CK(hipMemset(sums, 0, 64));
        if (aos) hipLaunchKernelGGL(k_verify<true>, dim3(4096), dim3(256), map_bytes(rows), 0, packed, dm, leafbase, s, okey, oidx, orec, sums, sums + 4);
        else hipLaunchKernelGGL(k_verify<false>, dim3(4096), dim3(256), map_bytes(rows), 0, packed, dm, leafbase, s, okey, oidx, orec, sums, sums + 4);
        unsigned long long got[7]; CK(hipMemcpy(got, sums, 56, hipMemcpyDeviceToHost));
        const bool ok = got[0] == want[0] && got[1] == want[1] && got[4] == 0;
        printf("%-44s best %7.3f ms  avg %7.3f ms  alg %6.0f GB/s (%.3f of 8 TB/s)  phys %6.0f GB/s  %s\n", name, best, avg,
               alg / best / 1e6, alg / best / 1e6 / 8000.0, phys / best / 1e6, ok ? "ok" : "MISMATCH");
        if (!ok) printf("    sum idx %llu vs %llu, sq %llu vs %llu, wrong bucket %llu, wrong key %llu\n", got[0], want[0], got[1], want[1], got[5], got[6]);
        fflush(stdout);
    };
#define RUN(NAME, AOSV, LDS, KERNEL, NTHR, GRIDX)                                                                         \
    if (strstr(NAME, only)) {                                                                                             \
        float best = 1e9f, tot = 0; const int reps = 4;                                                                   \
        CK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)));             \
        for (int r = 0; r < reps; r++) {                                                                                  \
            CK(hipMemcpy(gcur, gcur0, (size_t)G * NB * 4, hipMemcpyDeviceToDevice));                                      \
            CK(hipEventRecord(e0, 0));                                                                                    \
            hipLaunchKernelGGL(KERNEL, dim3((uint32_t)(GRIDX)), dim3(NTHR), LDS, 0, n, packed, dm, chunk, G, gcur, okey, oidx, orec); \
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());                                \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) { tot += ms; if (ms < best) best = ms; }               \
        }                                                                                                                 \
        report(NAME, AOSV, best, tot / (reps - 1));                                                                       \
    }
    // reference streams: 18 GB fill, 18 GB copy
    {
        const uint64_t nvec = s * 12 / 16;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (uint4*)orec, nvec);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 2) printf("fill %.1f GB: %.3f ms = %.0f GB/s\n", nvec * 16 / 1e9, ms, nvec * 16 / ms / 1e6);
        }
        const uint64_t nv2 = s * 8 / 16;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4*)orec, (uint4*)okey, nv2);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 2) printf("copy %.1f GB -> %.1f GB: %.3f ms = %.0f GB/s (read + write)\n", nv2 * 16 / 1e9, nv2 * 16 / 1e9, ms, 2.0 * nv2 * 16 / ms / 1e6);
        }
    }
    const size_t lds_st = (size_t)4864 * 12 + map_bytes(rows) + (size_t)((NB + 3) & ~3u) * 4 + 128;
    const size_t lds_d = map_bytes(rows) + (size_t)((NB + 3) & ~3u) * 8;
#define LDS_ST(CAP_) ((size_t)(CAP_) * 12 + map_bytes(rows) + (size_t)((NB + 3) & ~3u) * 4 + 128)
    RUN("staged SoA 512x16 (today)", false, LDS_ST(4864), (k_staged<false, 512, 16, 4864, 4, 0>), 512, nwg)
    RUN("staged AoS 512x16 cap4864", true, LDS_ST(4864), (k_staged<true, 512, 16, 4864, 4, 0>), 512, nwg)
    RUN("staged AoS 256x16 cap2432", true, LDS_ST(2432), (k_staged<true, 256, 16, 2432, 4, 0>), 256, nwg)
    RUN("staged AoS 1024x16 cap9728", true, LDS_ST(9728), (k_staged<true, 1024, 16, 9728, 4, 0>), 1024, nwg)
    RUN("staged AoS 1024x8 cap4864", true, LDS_ST(4864), (k_staged<true, 1024, 8, 4864, 4, 0>), 1024, nwg)
    RUN("staged AoS 512x16 nostore", true, LDS_ST(4864), (k_staged<true, 512, 16, 4864, 4, 1>), 512, nwg)
    RUN("staged AoS 512x16 densestore", true, LDS_ST(4864), (k_staged<true, 512, 16, 4864, 4, 2>), 512, nwg)
    RUN("staged SoA 512x16 densestore", false, LDS_ST(4864), (k_staged<false, 512, 16, 4864, 4, 2>), 512, nwg)
#define LDS_BT(RNDS_, NT_, CAP_) ((size_t)((NT_) * 32 * (RNDS_) * 3 / 32 + 4) * 4 + map_bytes(rows) + (size_t)((NB + 3) & ~3u) * 8 + 128 + (size_t)(CAP_) * 2)
    RUN("bigtile 512x2x32 cap18432", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4>), 512, nwg)
    RUN("bigtile2 512x2x32 sb8", true, LDS_BT(2, 512, 18432), (k_bigtile2<512, 2, 18432, 4, 8>), 512, nwg)
    RUN("bigtile2 512x2x32 sb4", true, LDS_BT(2, 512, 18432), (k_bigtile2<512, 2, 18432, 4, 4>), 512, nwg)
    RUN("bigtile2 512x2x32 sb0", true, LDS_BT(2, 512, 18432), (k_bigtile2<512, 2, 18432, 4, 0>), 512, nwg)
    RUN("bigtile2 1024x1x32 sb8", true, LDS_BT(1, 1024, 18432), (k_bigtile2<1024, 1, 18432, 4, 8>), 1024, nwg)
    RUN("bigtile2 1024x2x32 sb8", true, LDS_BT(2, 1024, 36864), (k_bigtile2<1024, 2, 36864, 4, 8>), 1024, nwg)
    RUN("bigtile 512x2x32 abl5 4B stores", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 5>), 512, nwg)
    RUN("bigtile 512x2x32 abl6 16B stores", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 6>), 512, nwg)
    RUN("bigtile 512x2x32 abl7 dense stores", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 7>), 512, nwg)
    RUN("bigtile 512x2x32 abl8 SoA stores", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 8>), 512, nwg)
    RUN("bigtile 512x2x32 stag>>8", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 0, 8>), 512, nwg)
    RUN("bigtile 512x2x32 stag>>3", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 0, 3>), 512, nwg)
    RUN("bigtile 512x2x32 stag>>4", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 0, 4>), 512, nwg)
    RUN("bigtile 1024x2x32 stag>>3", true, LDS_BT(2, 1024, 36864), (k_bigtile<1024, 2, 36864, 4, 0, 3>), 1024, nwg)
    RUN("bigtile 1024x2x32 stag>>4", true, LDS_BT(2, 1024, 36864), (k_bigtile<1024, 2, 36864, 4, 0, 4>), 1024, nwg)
    RUN("bigtile 512x2x32 abl1 nostore", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 1>), 512, nwg)
    RUN("bigtile 512x2x32 abl2 no copy-out", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 2>), 512, nwg)
    RUN("bigtile 512x2x32 abl3 no pass B", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 3>), 512, nwg)
    RUN("bigtile 512x2x32 abl4 no pass A", true, LDS_BT(2, 512, 18432), (k_bigtile<512, 2, 18432, 4, 4>), 512, nwg)
    RUN("bigtile 512x4x32 cap36864", true, LDS_BT(4, 512, 36864), (k_bigtile<512, 4, 36864, 4>), 512, nwg)
    RUN("bigtile 1024x2x32 cap36864", true, LDS_BT(2, 1024, 36864), (k_bigtile<1024, 2, 36864, 4>), 1024, nwg)
    RUN("bigtile 256x2x32 cap9216", true, LDS_BT(2, 256, 9216), (k_bigtile<256, 2, 9216, 4>), 256, nwg)
    return 0;
}
