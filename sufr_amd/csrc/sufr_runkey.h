// sufr_runkey.h -- key formats shared by the device kernels and the host-side unit tests.
// Everything here is plain integer code on a padded text; SUFR_HD makes it callable from both sides.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SUFR_HD __host__ __device__ __forceinline__
#else
#define SUFR_HD static inline
#endif

namespace sufr {

static constexpr uint32_t RUN_SAT = (1u << 29) - 1u;   // saturation of plain run lengths inside run keys: token <= 60 bits,
                                                       // which five 12-bit passes of a re-keying level can still sort
static constexpr uint32_t PERIOD_SAT = 65535u;         // saturation of periodic extensions (found by a word-wise scan)
static constexpr uint32_t RUN_TILE = 4096u;   // granularity of the run-end table (== TILE of the kernels)
static constexpr uint32_t RUN_NONE = 0xffffffffu;

// Run ends of the normalised text, written by the normalise pass (p is a run end iff p == n-1 or
// text[p] != text[p+1]), at three granularities:
//   ends[w]      bit i set: position 64 w + i is a run end                  (n / 8 bytes)
//   tile_any[t]  bit j set: ends[64 t + j] != 0                             (8 bytes per 4096-byte tile)
//   first_end[t] first run end of tile t, RUN_NONE if there is none         (4 bytes per tile)
//   next_tile[t] smallest t' >= t whose tile holds a run end                (4 bytes per tile)
// so the length of a run is found in a handful of loads whether it is 3 bytes or 30 million.
struct RunTable {
    const uint64_t* ends;
    const uint64_t* tile_any;
    const uint32_t* first_end;
    const uint32_t* next_tile;
    uint32_t ntiles;
};

// unaligned 8-byte little-endian load (the text buffer is padded with >= 64 zero bytes)
SUFR_HD uint64_t load_u64_unaligned(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// x / bits for the handful of character widths that occur (bits is uniform across the wave, so this is a
// scalar branch plus a multiply-shift instead of a software division)
SUFR_HD uint32_t div_by_bits(uint32_t x, int bits)
{
    switch (bits) {
        case 1: return x;
        case 2: return x >> 1;
        case 3: return x / 3u;
        case 4: return x >> 2;
        case 5: return x / 5u;
        case 6: return x / 6u;
        case 7: return x / 7u;
        case 8: return x >> 3;
        default: return x / 9u;
    }
}

// common leading characters of two plain packed keys (K characters of `bits` bits from the top)
SUFR_HD uint32_t plain_key_common(uint64_t a, uint64_t b, int bits, int K)
{
    uint64_t x = a ^ b;
    return x ? div_by_bits((uint32_t)__builtin_clzll(x), bits) : (uint32_t)K;
}

// ---------------------------------------------------------------------------------------------
// Run keys: the key format of every level below the first and of every tie round of the finisher.
// All suffixes of a group agree on their first `q - idx` characters, in particular on c = text[q-1].
// The key describes the text from q on as
//     cls | gamma(rem + 1) [complemented if cls] | the characters after the run, b bits each
//   rem = number of further bytes equal to c starting at q (0 if text[q] != c), saturating at RUN_SAT;
//   x   = the byte after that run (end of text sorts lowest);   cls = (x > c).
// Comparing  c^remA xA...  with  c^remB xB...:  if the runs differ in length, the shorter one is
// smaller iff its x is below c -- so  {x < c, rem ascending} < {x > c, rem descending}; the Elias-gamma
// code (L ones, a zero, L low bits) is order preserving and its complement reverses the order.
// A saturated run has x == c, cls = 0 and continues with real text characters, which keeps the order.
// The integer order of run keys therefore equals the suffix order, and a run of any length costs one
// run-table lookup instead of a byte-by-byte walk.
// ---------------------------------------------------------------------------------------------
struct RunTok { uint32_t cls, rem; int tokbits; };

SUFR_HD RunTok decode_run_token(uint64_t key)
{
    RunTok t;
    t.cls = (uint32_t)(key >> 63);
    uint64_t g = key << 1;
    if (t.cls) g = ~g;
    int L = g == ~0ull ? 63 : __builtin_clzll(~g);    // leading ones
    if (L > 29) L = 29;                        // rem + 1 <= 2^29
    uint32_t low = L ? (uint32_t)((g << (L + 1)) >> (64 - L)) : 0u;
    t.rem = ((1u << L) | low) - 1u;
    t.tokbits = 2 + 2 * L;
    return t;
}

// characters of common prefix described by two DIFFERENT keys of one group
SUFR_HD uint32_t run_key_common(uint64_t a, uint64_t b, int bits)
{
    RunTok ta = decode_run_token(a), tb = decode_run_token(b);
    if (ta.cls != tb.cls || ta.rem != tb.rem) return (ta.rem < tb.rem ? ta.rem : tb.rem);
    uint64_t x = (a ^ b) << ta.tokbits;
    return ta.rem + (x ? div_by_bits((uint32_t)__builtin_clzll(x), bits) : div_by_bits((uint32_t)(64 - ta.tokbits), bits));
}

// characters covered by the top `sorted_bits` of a run key (what a group defined on them shares)
SUFR_HD uint32_t run_key_advance(uint64_t key, int sorted_bits, int bits)
{
    RunTok t = decode_run_token(key);
    int plain = sorted_bits - t.tokbits;
    if (plain < 0) return 0u;          // the token itself is not fully shared: nothing is known to agree
    return t.rem + (plain > 0 ? div_by_bits((uint32_t)plain, bits) : 0u);
}

// periodic extension length for period pi > 1: number of bytes from q on that equal the byte pi
// positions earlier, capped at PERIOD_SAT (word-wise scan; tandem arrays are kilobases, not megabases)
SUFR_HD uint32_t periodic_rem(const uint8_t* __restrict__ text, uint64_t n, uint64_t q,
                                                 uint32_t pi)
{
    uint32_t rem = 0;
    while (rem < PERIOD_SAT && q + rem < n) {
        uint64_t a = load_u64_unaligned(text + q + rem);
        uint64_t b = load_u64_unaligned(text + q + rem - pi);
        uint64_t x = a ^ b;
        uint32_t same = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
        uint64_t room = n - (q + rem);
        if (same > room) same = (uint32_t)room;
        rem += same;
        if (same < 8) break;
    }
    return rem < PERIOD_SAT ? rem : PERIOD_SAT;
}

// run_len_exact: length of the run of equal bytes that starts at q, q < n; runs end at the end of the text.  run_len_at: the same,
// saturating at RUN_SAT (what a run key can hold).
// Genomes built with --ignore-softmask are ~50 % 'N' in runs of hundreds to millions of bytes, and the
// reference walks through them byte by byte inside find_lcp (sufr_builder.rs:301-331).  Here a run of any
// length costs one bitmap word, at most two more loads inside its 4 KB tile, and two for all tiles after it.
SUFR_HD uint64_t run_len_exact(uint64_t q, RunTable rt)
{
    uint64_t w = q >> 6;
    uint64_t m = rt.ends[w] & (~0ull << (q & 63u));
    if (!m) {
        const uint64_t tile = q / RUN_TILE;
        const uint32_t j = (uint32_t)(w & 63u);
        const uint64_t later = j == 63u ? 0ull : (rt.tile_any[tile] & (~0ull << (j + 1u)));
        if (later) {
            w = (tile << 6) + (uint64_t)__builtin_ctzll(later);
            m = rt.ends[w];
        } else {
            if (tile + 1 >= rt.ntiles) return ~0ull;                       // cannot happen: n-1 is a run end
            const uint32_t fe = rt.first_end[rt.next_tile[tile + 1]];
            return (uint64_t)fe - q + 1;
        }
    }
    return (w << 6) + (uint64_t)__builtin_ctzll(m) - q + 1;
}
SUFR_HD uint32_t run_len_at(uint64_t q, RunTable rt)
{
    const uint64_t len = run_len_exact(q, rt);
    return len < RUN_SAT ? (uint32_t)len : RUN_SAT;
}

// lowest bit of every whole `bits`-bit code of a 64-bit word filled from the top (code j in bits [64 - bits (j + 1), 64 - bits j))
SUFR_HD uint64_t code_lsb_mask(int bits)
{
    switch (bits) {
        case 1: return ~0ull;
        case 2: return ~0ull / 3u;
        case 3: return ~0ull / 7u;
        case 4: return ~0ull / 15u;
        case 5: return ~0ull / 31u;
        case 6: return ~0ull / 63u;
        case 7: return ~0ull / 127u;
        case 8: return ~0ull / 255u;
        default: return ~0ull / 511u;
    }
}

// number of leading whole codes of `v` (codes from the top) that equal `code`; 64 / bits when all of them do
SUFR_HD uint32_t leading_equal_codes(uint64_t v, uint32_t code, int bits)
{
    const uint32_t K = div_by_bits(64u, bits);
    const int spare = 64 - (int)K * bits;                      // low bits that hold no whole code
    const uint64_t x = (v ^ ((uint64_t)code * code_lsb_mask(bits))) >> spare;
    return x ? div_by_bits((uint32_t)__builtin_clzll(x << spare), bits) : K;
}

// The run key of position q from the packed code stream alone (period 1), in two steps so that a caller with several
// positions in hand can issue all their first reads before it waits for any (k_finish re-keys two slots per lane: one
// random-sector latency per round instead of two): packed_fetch16(q - 1) = the 16 bytes that hold the codes at q - 1, q and
// the ~20 after them; make_run_key_packed() turns them into the key.  Codes are the ranks of the bytes: equal codes <=> equal
// bytes, code order = byte order.  Only a run (text[q] == text[q - 1]) that does not end inside the codes in hand costs more:
// the run-end table, then the same read behind the run.
struct PackedPair { uint64_t hi, lo; };
SUFR_HD PackedPair packed_fetch16(const uint8_t* __restrict__ packed, int bits, uint64_t p)
{
    const uint8_t* pp = packed + ((p * (uint64_t)bits) >> 3);
    PackedPair w;
    w.hi = __builtin_bswap64(load_u64_unaligned(pp)); w.lo = __builtin_bswap64(load_u64_unaligned(pp + 8));
    return w;
}
// 64 bits of codes from p and from p + 1 on, out of the 16 bytes fetched for p
SUFR_HD void packed_codes_at(PackedPair w, int bits, uint64_t p, uint64_t& from_p, uint64_t& from_next)
{
    const uint32_t s0 = (uint32_t)((p * (uint64_t)bits) & 7u), s1 = s0 + (uint32_t)bits;        // s1 <= 11
    from_p = s0 ? ((w.hi << s0) | (w.lo >> (64 - s0))) : w.hi;
    from_next = (w.hi << s1) | (w.lo >> (64 - s1));
}
SUFR_HD uint64_t make_run_key_packed(uint64_t n, RunTable rt, int bits, uint64_t q, const uint8_t* __restrict__ packed,
                                     PackedPair first /* packed_fetch16(packed, bits, q - 1) */)
{
    uint32_t rem = 0;
    uint64_t vprev, vq;
    packed_codes_at(first, bits, q - 1, vprev, vq);
    uint32_t cprev = (uint32_t)(vprev >> (64 - bits)), cq = (uint32_t)(vq >> (64 - bits));
    if (q < n && cq == cprev) {
        // a run that ends inside the ~20 codes in hand (every fourth suffix of a DNA text continues its last character;
        // almost all such runs are a few characters long) needs no look at the run-end table: another random sector
        const uint32_t eq = leading_equal_codes(vq, cprev, bits);
        rem = eq < div_by_bits(64u, bits) ? eq : run_len_at(q, rt);
        packed_codes_at(packed_fetch16(packed, bits, q + rem - 1), bits, q + rem - 1, vprev, vq);
        cprev = (uint32_t)(vprev >> (64 - bits)); cq = (uint32_t)(vq >> (64 - bits));
    }
    const uint64_t after = q + rem;
    const uint32_t cls = (after < n && cq > cprev) ? 1u : 0u;
    const uint32_t v = rem + 1u;
    const int L = 31 - __builtin_clz(v);
    const int glen = 2 * L + 1;
    uint64_t g = (((1ull << L) - 1ull) << (L + 1)) | (uint64_t)(v & ((1u << L) - 1u));
    if (cls) g = ~g & ((1ull << glen) - 1ull);
    const int tokbits = 1 + glen;
    const uint64_t key = ((uint64_t)cls << 63) | (g << (63 - glen));
    const int shift = 64 - tokbits;
    const int nch = (int)div_by_bits((uint32_t)shift, bits);
    const int spare = shift - nch * bits;                                        // unused low bits
    return key | ((vq >> tokbits) & (~0ull << spare));                           // vq: the codes from `after` on
}

// pi = period assumed for the group (1 = plain runs, served by the run-end table).  Any pi <= the length of
// the group's common prefix gives a valid order: all members agree on the pi bytes before q, hence on the
// periodic extension up to the shorter of their two break points, and at the break the suffix whose text
// leaves the extension is smaller iff its byte is below the byte the extension predicts there.
SUFR_HD uint64_t make_run_key(const uint8_t* __restrict__ text, uint64_t n,
                                                 RunTable rt, const uint16_t* s_lut,
                                                 int bits, uint64_t q, uint32_t pi,
                                                 const uint8_t* __restrict__ packed = nullptr)
{
    // q >= pi >= 1 and q <= n.  The text is padded, so loads next to q are issued without bounds checks.
    uint32_t rem = 0;
    uint64_t pw = 0; uint32_t pw2 = 0; uint32_t ps = 0;
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    uint64_t after;
    uint32_t x, c;
    if (packed && pi == 1) return make_run_key_packed(n, rt, bits, q, packed, packed_fetch16(packed, bits, q - 1));
    {
        if (pi == 1) {
            const uint32_t cprev = text[q - 1], cq = text[q];
            if (q < n && cq == cprev) rem = run_len_at(q, rt);
        } else {
            rem = periodic_rem(text, n, q, pi);
        }
        after = q + rem;                               // <= n
        const uint8_t* ta = text + after;
        x = ta[0];
        c = ta[-(int)pi];                              // what the periodic extension predicts at `after`
        // `packed` (optional): the whole text as a big-endian stream of `bits`-bit codes, zero past the end;
        // the characters after the run are then one unaligned 9-byte read instead of a per-character loop
        if (packed) {
            const uint64_t o = after * (uint64_t)bits;
            const uint8_t* pp = packed + (o >> 3);
            ps = (uint32_t)(o & 7u);
            pw = __builtin_bswap64(load_u64_unaligned(pp));
            pw2 = pp[8];
        } else {
            w0 = load_u64_unaligned(ta); w1 = load_u64_unaligned(ta + 8);
            w2 = load_u64_unaligned(ta + 16); w3 = load_u64_unaligned(ta + 24);
        }
    }
    const uint8_t* ta = text + after;
    const uint32_t cls = (after < n && x > c) ? 1u : 0u;
    const uint32_t v = rem + 1u;
    const int L = 31 - __builtin_clz(v);
    const int glen = 2 * L + 1;
    uint64_t g = (((1ull << L) - 1ull) << (L + 1)) | (uint64_t)(v & ((1u << L) - 1u));
    if (cls) g = ~g & ((1ull << glen) - 1ull);
    const int tokbits = 1 + glen;
    uint64_t key = ((uint64_t)cls << 63) | (g << (63 - glen));
    int shift = 64 - tokbits;
    const int nch = (int)div_by_bits((uint32_t)shift, bits);
    if (packed) {
        const uint64_t v = ps ? ((pw << ps) | ((uint64_t)pw2 >> (8 - ps))) : pw;   // codes from `after` on
        const int spare = shift - nch * bits;                                        // unused low bits
        return key | ((v >> tokbits) & (~0ull << spare));
    }
    const uint64_t room = n - after;                   // characters that exist from `after` on
    const int live = room < (uint64_t)nch ? (int)room : nch;
    for (int j = 0; j < live; j++) {
        uint64_t w = j < 8 ? w0 : (j < 16 ? w1 : (j < 24 ? w2 : (j < 32 ? w3 : load_u64_unaligned(ta + (j & ~7)))));
        uint32_t code = (uint32_t)(s_lut[(w >> (8 * (j & 7))) & 0xffu] & 0x3ffu);
        shift -= bits;
        key |= (uint64_t)code << shift;
    }
    return key;
}


// Counting digit of the leaf sort for keys of the fixed 3-bit DNA codes (0 pad, 1 '$', 2 '%', 3 A, 4 C, 5 G, 6 N,
// 7 T): the six characters in the low 18 bits of t (first character highest) go to 2 bits each, A C G T -> 0 1 2 3.
// The other codes start no interval of their own: pad '$' '%' sit at the left end of A's, N at the left end of
// T's, and NOTHING AFTER such a character counts (zeros) -- a point, not an interval.  That keeps the digest
// monotone: t < t'  =>  digest(t) <= digest(t'), which is all the leaf sort needs of it (merging two characters
// into one value without cutting the tail would not be: "GT" < "NA" but 2 3 > 2 0).
SUFR_HD uint32_t dna3_digest12(uint32_t t)
{
    const uint32_t Q = 0x24924u;                                  // bit 2 of every 3-bit field
    const uint32_t B2 = t & Q, B1 = (t << 1) & Q, B0 = (t << 2) & Q;
    const uint32_t M1 = B2 & (B1 | B0);                           // codes 5 6 7
    const uint32_t M0 = B2 & (B1 | (B0 ^ Q));                     // codes 4 6 7
    const uint32_t ST = (~B2 & ~(B1 & B0) & Q) | (B2 & B1 & (B0 ^ Q));   // codes 0 1 2 and 6: the tail is cut
    uint32_t W = (M1 | (M0 >> 1)) >> 1;                           // character i: value in bits [3i + 1 : 3i]
    uint32_t S = ST >> 2;                                         // character i: cut flag in bit 3i
    S |= S >> 3; S |= S >> 6; S |= S >> 12;                       // every character at or after a cut
    const uint32_t Z = S >> 3;                                    // strictly after
    W &= ~(Z | (Z << 1));
    const uint32_t P = (W & 0x030c3u) | ((W >> 1) & 0x0c30cu);    // pairs of characters: 4 bits at 0, 6, 12
    return (P & 0xfu) | ((P >> 2) & 0xf0u) | ((P >> 4) & 0xf00u);
}

}  // namespace sufr
