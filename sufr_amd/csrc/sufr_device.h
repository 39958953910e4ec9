// sufr_device.h -- constants and POD parameter blocks shared by the kernels and the host pipeline.
#pragma once
#include <stdint.h>

namespace sufr {

static constexpr int THREADS = 256;          // workgroup size of the radix kernels (4 wavefronts)
static constexpr int EPT = 16;               // records / text positions per thread and tile
static constexpr int TILE = THREADS * EPT;   // 4096
static constexpr int HALO = 128;             // >= max K (64); text tile over-read for the rolling key
static constexpr int TEXT_PAD = 2 * TILE + HALO + 64;  // the largest scatter tile is 8192  // zero bytes after the text in the workspace copy

// Key layout for one build (derived from the byte histogram of the normalised text).
struct KeyParams {
    int b;          // bits per character code (codes 1..sigma, 0 = past end of text)
    int K;          // characters per 64-bit key  = 64 / b
    int dchars;     // characters per radix digit = max(1, 12 / b)
    int dbits;      // bits per radix digit       = b * dchars  (<= 12)
    uint32_t raw_bins; // 1 << dbits: nominal digit values
    uint32_t nbins;    // digit values after the dense remap (== raw_bins when there is no remap)
    int top_shift;  // 64 - dbits: the most significant digit (shard selector)
    const uint8_t* packed; // text as a big-endian stream of b-bit codes (b <= 4), or nullptr
    uint32_t elig_codes;   // bit c set: a suffix may start with the character whose code is c (b <= 4)
    int detect_period; // deep levels: use periodic run tokens for groups whose common prefix is periodic
};

// A record of the MSD levels: the packed K-character key of a suffix and its position, 12 bytes in ONE array.
// With the key and the index in separate arrays every digit run of a tile is two short stores to two far-apart
// lines; as one array of records it is one store of 1.5 times the length (the partition kernel: 13.3 -> 9.7 ms
// on the probe of profiles/micro/part_bench.hip, same staging).
struct Rec {
    uint32_t klo, khi, idx;
};

}  // namespace sufr
