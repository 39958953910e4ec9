// Host-side shim around sufr_amd/csrc/sufr_runkey.h (the key formats the device kernels use), so that the
// order-preservation and LCP-decoding properties can be tested on CPU.  Test infrastructure.
#include <stdint.h>
#include <stdlib.h>
#include "../sufr_amd/csrc/sufr_runkey.h"

extern "C" {

// reference for run_len_at: min(RUN_SAT, run of equal bytes starting at p, ending at n), one byte at a time
void shim_run_lengths(const uint8_t* text, uint64_t n, uint32_t* R)
{
    uint64_t p = n;
    uint32_t run = 0;
    while (p-- > 0) {
        if (p + 1 < n && text[p + 1] == text[p]) run = run < sufr::RUN_SAT ? run + 1 : sufr::RUN_SAT;
        else run = 1;
        R[p] = run;
    }
}
// the run-end tables k_normalize_bytehist writes (RunTable of sufr_runkey.h); ends: 64 words per tile
void shim_run_table(const uint8_t* text, uint64_t n, uint64_t* ends, uint64_t* tile_any, uint32_t* first_end)
{
    const uint64_t ntiles = (n + sufr::RUN_TILE - 1) / sufr::RUN_TILE;
    for (uint64_t t = 0; t < ntiles; t++) {
        first_end[t] = sufr::RUN_NONE;
        tile_any[t] = 0;
        for (int j = 0; j < 64; j++) ends[t * 64 + j] = 0;
        for (uint64_t p = t * sufr::RUN_TILE; p < n && p < (t + 1) * sufr::RUN_TILE; p++)
            if (p == n - 1 || text[p] != text[p + 1]) {
                if (first_end[t] == sufr::RUN_NONE) first_end[t] = (uint32_t)p;
                ends[p >> 6] |= 1ull << (p & 63);
                tile_any[t] |= 1ull << ((p >> 6) & 63);
            }
    }
}
static sufr::RunTable shim_table(uint64_t n, const uint64_t* ends)
{
    // layout used by the tests: [ends: 64 words per tile][tile_any: 1 word per tile][first_end: u32 per tile]
    //                           [next_tile: u32 per tile]
    const uint64_t ntiles = (n + sufr::RUN_TILE - 1) / sufr::RUN_TILE;
    const uint32_t* fe = (const uint32_t*)(ends + ntiles * 65);
    return sufr::RunTable{ends, ends + ntiles * 64, fe, fe + ntiles, (uint32_t)ntiles};
}
void shim_run_table_packed(const uint8_t* text, uint64_t n, uint64_t* buf)
{
    const uint64_t ntiles = (n + sufr::RUN_TILE - 1) / sufr::RUN_TILE;
    uint32_t* fe = (uint32_t*)(buf + ntiles * 65);
    shim_run_table(text, n, buf, buf + ntiles * 64, fe);
    uint32_t* nt = fe + ntiles;                           // what the k_next_tile_* kernels compute
    uint32_t nxt = sufr::RUN_NONE;
    for (uint64_t t = ntiles; t-- > 0;) {
        if (fe[t] != sufr::RUN_NONE) nxt = (uint32_t)t;
        nt[t] = nxt;
    }
}
uint32_t shim_run_len_at(uint64_t n, uint64_t q, const uint64_t* tab)
{
    return sufr::run_len_at(q, shim_table(n, tab));
}
uint64_t shim_make_run_key(const uint8_t* text, uint64_t n, const uint64_t* tab, const uint16_t* lut, int bits,
                           uint64_t q, uint32_t pi, const uint8_t* packed)
{
    return sufr::make_run_key(text, n, shim_table(n, tab), lut, bits, q, pi, packed);
}
// what k_pack_codes produces: big-endian stream of `bits`-bit codes, zero past the end
void shim_pack_codes(const uint8_t* text, uint64_t n, const uint16_t* lut, int bits, uint8_t* packed, uint64_t cap)
{
    for (uint64_t i = 0; i < cap; i++) packed[i] = 0;
    for (uint64_t p = 0; p < n; p++) {
        uint32_t c = lut[text[p]] & 0x7fu;
        for (int k = 0; k < bits; k++) {
            uint64_t bit = p * bits + k;                        // bit index from the MSB of byte 0
            if ((c >> (bits - 1 - k)) & 1u) packed[bit >> 3] |= (uint8_t)(0x80u >> (bit & 7));
        }
    }
}
uint32_t shim_run_key_common(uint64_t a, uint64_t b, int bits) { return sufr::run_key_common(a, b, bits); }
uint32_t shim_run_key_advance(uint64_t k, int sorted_bits, int bits) { return sufr::run_key_advance(k, sorted_bits, bits); }
uint32_t shim_plain_key_common(uint64_t a, uint64_t b, int bits, int K) { return sufr::plain_key_common(a, b, bits, K); }
// dna3_digest12 over all 2^18 six-character strings: out[t]
void shim_dna3_digest_all(uint32_t* out)
{
    for (uint32_t t = 0; t < (1u << 18); t++) out[t] = sufr::dna3_digest12(t);
}
}
