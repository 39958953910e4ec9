// Host-side shim around sufr_amd/csrc/sufr_runkey.h (the key formats the device kernels use), so that the
// order-preservation and LCP-decoding properties can be tested on CPU.  Test infrastructure.
#include <stdint.h>
#include <stdlib.h>
#include "../sufr_amd/csrc/sufr_runkey.h"

extern "C" {

// R[p] = min(65535, run of equal bytes starting at p, ending at n); what k_run_first/k_run_fill compute
void shim_run_lengths(const uint8_t* text, uint64_t n, uint16_t* R)
{
    uint64_t p = n;
    uint32_t run = 0;
    while (p-- > 0) {
        if (p + 1 < n && text[p + 1] == text[p]) run = run < sufr::RUN_SAT ? run + 1 : sufr::RUN_SAT;
        else run = 1;
        R[p] = (uint16_t)run;
    }
}
uint64_t shim_make_run_key(const uint8_t* text, uint64_t n, const uint16_t* R, const uint16_t* lut, int bits,
                           uint64_t q, uint32_t pi, const uint8_t* packed)
{
    return sufr::make_run_key(text, n, R, lut, bits, q, pi, packed);
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
}
