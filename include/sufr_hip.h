/*
 * sufr_hip.h -- C ABI of libsufr_hip.so: MI355X-native suffix-array + LCP construction.
 *
 * Drop-in boundary for the construction path of TravisWheelerLab/sufr (v0.7.12).  The reference has
 * no FFI of its own (it is 100 % Rust); the seam is
 *     sufr::create                      sufr/src/lib.rs:321-371
 *      -> SuffixArray::write(args)      libsufr/src/suffix_array.rs:460-470   (u32 / u64 width rule)
 *      -> SufrBuilder::<T>::new(args)   libsufr/src/sufr_builder.rs:143-220
 *           normalise text (144-160) -> sort() (495-598) -> write() (817-918)
 * Each entry point below names the reference item it replaces.  INTEGRATION.md shows the Rust
 * `extern "C"` binding a maintainer would add to libsufr.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on success or
 * a negative SUFR_HIP_E_* code and never throws; the message for the last failure of a context is
 * sufr_hip_last_error(ctx).  A context is bound to one GPU and must be used from one thread at a
 * time; the library installs no signal handlers and keeps no global state.
 */
#ifndef SUFR_HIP_H
#define SUFR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUFR_HIP_ABI_VERSION 3

/* .sufr serialisation version, libsufr/src/types.rs:16 */
#define SUFR_OUTFILE_VERSION 6
/* sentinel appended to the text by read_sequence_file, libsufr/src/types.rs:20 */
#define SUFR_SENTINEL_CHARACTER '$'

/* flags: the boolean fields of SufrBuilderArgs, libsufr/src/types.rs:527-582 */
#define SUFR_HIP_FLAG_DNA              (1u << 0) /* is_dna */
#define SUFR_HIP_FLAG_ALLOW_AMBIGUITY  (1u << 1) /* allow_ambiguity */
#define SUFR_HIP_FLAG_IGNORE_SOFTMASK  (1u << 2) /* ignore_softmask (used only with RAW_TEXT) */
#define SUFR_HIP_FLAG_RAW_TEXT         (1u << 3) /* input is not yet normalised: apply the text map
                                                    of sufr_builder.rs:144-160 on the GPU */

/* error codes */
#define SUFR_HIP_OK              0
#define SUFR_HIP_E_INVALID      -1  /* bad argument */
#define SUFR_HIP_E_NO_DEVICE    -2  /* no usable HIP device: the library never falls back to CPU */
#define SUFR_HIP_E_HIP          -3  /* HIP runtime error (message has the call) */
#define SUFR_HIP_E_NOMEM        -4
#define SUFR_HIP_E_CAPACITY     -5  /* output buffers too small for num_suffixes */
#define SUFR_HIP_E_UNSUPPORTED  -6  /* valid reference option not built for the GPU yet */
#define SUFR_HIP_E_IO           -7
#define SUFR_HIP_E_CONFLICT     -8  /* "Cannot use max_query_len and seed_mask together" */
#define SUFR_HIP_E_SEED_MASK    -9  /* "Invalid seed mask '<mask>'" */

typedef struct sufr_hip_ctx sufr_hip_ctx;

/* Phase timings and sizes of the last build (replaces the reference's log::info! phase timers,
 * sufr_builder.rs:428,476,589).  Milliseconds are HIP-event times on the build stream. */
typedef struct sufr_hip_stats {
    uint64_t text_len;          /* n */
    uint64_t num_suffixes;      /* s (this shard) */
    uint32_t alphabet_size;     /* distinct bytes in the normalised text */
    uint32_t bits_per_char;     /* b */
    uint32_t chars_per_key;     /* K */
    uint32_t digit_bits;        /* radix digit width */
    uint32_t num_passes;        /* MSD levels of the top level (1 = the text partition only) */
    uint32_t num_levels;        /* 1 + re-keying levels needed for long repeats */
    uint64_t num_large_groups;  /* groups handed to deeper levels (all levels) */
    uint64_t deep_records;      /* records processed by deeper levels (all levels) */
    uint32_t top_lo, top_hi;    /* prefix-bucket range of this shard [lo, hi) */
    uint32_t partition_workgroups; /* grid of the radix-partition kernel */
    uint32_t partition_variant; /* 3 = k_msd_part_text (bit-packed stream, alphabets of <= 15 symbols), 4 = the same with half-size tiles
                                   (more than ~3 200 first digits), 0 = k_scatter_text (text staging) */
    float ms_total;             /* text resident in HBM -> SA+LCP resident in HBM */
    float ms_normalize;         /* k_text_pass_dna (DNA: normalise + pack + run ends + suffix-start bitmap + first-digit histogram, one pass) / k_normalize_bytehist */
    float ms_hist_text;         /* digit map + cursor setup of level 1 (the first-digit histogram itself is part of k_text_pass_dna; other
                                   alphabets: k_msd_hist_text / k_hist_text) */
    float ms_partition;         /* k_msd_part_text / k_scatter_text: THE radix-partition kernel (one launch) */
    float ms_passes;            /* further MSD levels + leaf sorts */
    float ms_finish;            /* boundary LCPs of the leaf windows + tie counts (k_fix_window_lcp, k_tie_counts) */
    float ms_deep;              /* everything below the MSD levels: left-over buckets and tie runs through the re-keying levels
                                   (k_gather_keys, k_group_sort_*, k_plan_windows, k_finish, prefix doubling) and the buckets of one
                                   repeated symbol placed by counting (sufr_runs.inc); their records are in deep_records only as far as
                                   they went through the levels */
    /* host phases (seconds).  sufr_hip_create_file: as named.  sufr_hip_build_u32 / _u64 (host buffers; round 6): host_read_s =
     * the caller's text to the device, host_build_s = device build (+ first-use allocations), host_write_s = text + SA + LCP
     * into the caller's buffers.  0 from the device-resident entry points. */
    float host_read_s;          /* sequence file -> text */
    float host_build_s;         /* H2D + device build (+ first-use allocations) */
    float host_write_s;         /* D2H + .sufr written */
    /* ABI 3 (round 6): DNA texts with a few bytes outside {$ % A C G N T} -- IUPAC codes, another delimiter -- keep the fixed 3-bit
     * code table: the build runs on 'N' in their place and the suffixes whose comparisons reached one are re-placed afterwards */
    uint32_t num_exceptions;    /* such bytes in the text (0: none, or the build took the general code table) */
    float ms_exceptions;        /* marking + re-placing (part of ms_deep and ms_total) */
    uint64_t num_reinserted;    /* suffixes re-placed by whole-text comparison */
} sufr_hip_stats;

/* ---- context ------------------------------------------------------------------------------- */
int  sufr_hip_abi_version(void);
/* number of HIP devices visible (0 when there is none; never initialises a context) */
int  sufr_hip_device_count(void);
/* replaces: rayon::ThreadPoolBuilder...build_global() (sufr/src/main.rs:32-40) as the place where
 * execution resources are chosen.  device_id: HIP ordinal.  Returns NULL on failure; the reason is
 * then available from sufr_hip_last_error(NULL). */
sufr_hip_ctx *sufr_hip_create(int device_id);
void sufr_hip_destroy(sufr_hip_ctx *ctx);
const char *sufr_hip_last_error(const sufr_hip_ctx *ctx);
/* optional: run the build on a caller-owned hipStream_t (e.g. torch's current stream) */
int  sufr_hip_set_stream(sufr_hip_ctx *ctx, void *hip_stream);
/* Waits for everything enqueued on the context's stream.  The context's own stream is non-blocking: work the caller
 * queued on another stream (the producer of a device text, the consumer of device results) is NOT ordered against it;
 * callers synchronise their producer before a call and use this after the calls that only enqueue. */
int  sufr_hip_synchronize(sufr_hip_ctx *ctx);
/* Texts of 2^32 - 2^24 bytes and more (the u64 arm of SuffixArray::write, suffix_array.rs:460-470) are built in
 * overlapping 32-bit windows that are merged by rank on the device (sufr_wide.inc).  window: positions per window,
 * margin: comparison context after them; 0, 0 selects the defaults (as few windows as fit, 2^26).  A non-zero window
 * also sends shorter texts of more than `window` bytes down the same path (memory-bound callers; the tests).
 * max_query_len / seed_mask builds take the same windows (built with the option, merged under its order).  A repeat
 * that crosses the end of a window and is longer than the margin makes the window re-build with the widest margin
 * (what 32 bits leave beside the window, ~2^31 symbols); one longer than that too is ordered by comparisons over the
 * whole text with 64-bit positions (round 5: the suffixes the window only saw a prefix of leave its arrays and are
 * merged back as a list of their own -- correct for any repeat, slow for long ones: every comparison walks the
 * repeat).  SUFR_HIP_E_UNSUPPORTED is left for more than 2^22 such suffixes in one window. */
int  sufr_hip_set_window(sufr_hip_ctx *ctx, uint64_t window, uint64_t margin);
/* widest_margin: cap of the margin of that re-build (0: what 32 bits allow).  A re-build with a 2^31-symbol margin is a
 * second full-size window: callers short of memory -- and the tests of the whole-text repair -- bound it. */
int  sufr_hip_set_window_retry(sufr_hip_ctx *ctx, uint64_t widest_margin);
/* suffixes of the context's last windowed build that were ordered by whole-text comparison (0: no window needed it) */
uint64_t sufr_hip_window_repairs(const sufr_hip_ctx *ctx);
/* Out-of-core form of a windowed `create` (sufr_hip_create_file / _from_sequence / _multi; the counterpart of the reference's
 * partitions sorted one at a time out of temporary files, sufr_builder.rs:495-598 + write() 875-906): `bytes` = device memory
 * the suffix and LCP arrays may occupy at once.  The build then runs in ceil(2 * n * width / bytes) shards (ranges of the first
 * 8 bytes), one after another on every context, and each shard's slice is streamed to its place in the file before the next is
 * built -- the whole arrays are never resident, on the device or the host.  0 (the default): no limit set; a create starts
 * with one shard per context and goes to 2, 4, 8 ... times as many while a build runs out of device memory.  The price is one windowed
 * sort of the text per shard.  Seed-mask builds and max_query_len < 8 are not sharded (they keep the whole-array path). */
int  sufr_hip_set_array_budget(sufr_hip_ctx *ctx, uint64_t bytes);

/* ---- text normalisation: sufr_builder.rs:144-160 (host helper; the GPU build can also do it) --- */
int sufr_hip_normalize(const uint8_t *in, uint8_t *out, uint64_t n, int ignore_softmask);

/* ---- the hot path: SufrBuilder::sort() + the SA/LCP stitch of write() ------------------------
 * replaces: partition() 404-487, sort() 495-598, merge_sort()/merge() 601-767, select_pivots()
 * 771-809 and the boundary-LCP fix 886-906 of libsufr/src/sufr_builder.rs.
 *
 * Device-resident variant (what bench.py times): d_text is a HIP device pointer to n text bytes,
 * d_sa / d_lcp are device arrays with room for `cap` entries.  With num_shards > 1 the call builds
 * only the shard_index-th prefix-bucket range (shards are balanced on the device from the k-mer
 * histogram; concatenating shards 0..num_shards-1 gives the full arrays, and the first LCP entry of
 * every shard but the first must be stitched: sufr_hip_stitch_device_u32 does it on the device under the order of the
 * build -- plain, seed mask or length cap --; sufr_hip_lcp_pair is the host form for plain builds).
 * max_query_len (0 = none) and seed_mask (NULL = none) are the reference's -m / -s builds
 * (sufr_builder.rs:272-300, 310-314, 350-359, 668-683): the seed-mask order (care characters, ties in
 * descending position, LCP in care characters) is reproduced exactly; for max_query_len, where the
 * reference's arrays depend on pivots and merge order, the canonical form is returned (order of the first
 * L characters, ties in descending position, LCP = min(exact, L)).  Both shard like plain
 * builds do ("a bounded key, then descending position": equal keys never straddle a first-digit boundary): a seed-mask
 * build on the first digit of its care-symbol key, a max_query_len build on the first digit of the plain order -- a cap
 * below 8 symbols (shorter than a digit can be) with num_shards > 1 returns SUFR_HIP_E_UNSUPPORTED.
 * num_partitions and random_seed are accepted for signature parity: the result does not depend on
 * them (pivots are replaced by on-device histogram splitters). */
int sufr_hip_sort_device_u32(sufr_hip_ctx *ctx, const void *d_text, uint64_t n, uint32_t flags,
                             uint64_t max_query_len, const char *seed_mask,
                             uint64_t num_partitions, uint64_t random_seed,
                             uint32_t shard_index, uint32_t num_shards,
                             void *d_sa, void *d_lcp, uint64_t cap,
                             uint64_t *num_suffixes_out, sufr_hip_stats *stats);
/* u64-index twin (SufrBuilder<u64>; suffix_array.rs:461 selects it when n >= u32::MAX).  Texts below
 * 2^32 - 2^24 bytes are built with 32-bit indices on the device and widened; longer texts take the windowed build
 * (sufr_hip_set_window above).  Round 5: a windowed build shards too (both entry points): the shards are ranges of the
 * suffixes' first 8 bytes, chosen from a fixed sample of the text -- identical on every rank, no communication --, every
 * window keeps the suffixes of the shard's range and the rank merge runs per shard; the shards concatenate to the single
 * build and the first LCP of a shard is stitched by sufr_hip_stitch_device_u32 / _u64.  (Every rank still builds every
 * window in full: what shrinks with the number of shards is the filter, the merge and the output.)  Seed-mask builds and
 * caps below 8 symbols are not sharded in windows (SUFR_HIP_E_UNSUPPORTED). */
int sufr_hip_sort_device_u64(sufr_hip_ctx *ctx, const void *d_text, uint64_t n, uint32_t flags,
                             uint64_t max_query_len, const char *seed_mask,
                             uint64_t num_partitions, uint64_t random_seed,
                             uint32_t shard_index, uint32_t num_shards,
                             void *d_sa, void *d_lcp, uint64_t cap,
                             uint64_t *num_suffixes_out, sufr_hip_stats *stats);

/* Boundary fix of a shard that stays in HBM (one shard per GPU; replaces the LCP write of the partition loop of
 * SufrBuilder::write, sufr_builder.rs:893-902, for device-resident arrays): d_bounds = [num_shards][3] uint64 in
 * device memory, {first suffix, last suffix, count} of every shard -- what the ranks all_gather after
 * sufr_hip_sort_device_u32 (24 bytes per rank).  Sets d_lcp[0] of shard `shard_index` to the exact LCP of its first
 * suffix with the last suffix of the nearest non-empty shard before it, on the device text of the context's last
 * build; the pair never visits the host.  Shard 0 and empty shards are left as they are. */
int sufr_hip_stitch_device_u32(sufr_hip_ctx *ctx, uint64_t n, const uint64_t *d_bounds, uint32_t shard_index,
                               uint32_t num_shards, void *d_lcp);
/* the same for a 64-bit LCP array (the shards of sufr_hip_sort_device_u64) */
int sufr_hip_stitch_device_u64(sufr_hip_ctx *ctx, uint64_t n, const uint64_t *d_bounds, uint32_t shard_index,
                               uint32_t num_shards, void *d_lcp);

/* Host-buffer variants: H2D copy of the text, build, D2H copy of SA and LCP.
 * If norm_text_out != NULL it receives the normalised text (what write() stores in the file). */
int sufr_hip_build_u32(sufr_hip_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t flags,
                       uint64_t max_query_len, const char *seed_mask,
                       uint64_t num_partitions, uint64_t random_seed,
                       uint8_t *norm_text_out, uint32_t *sa_out, uint32_t *lcp_out, uint64_t cap,
                       uint64_t *num_suffixes_out, sufr_hip_stats *stats);
int sufr_hip_build_u64(sufr_hip_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t flags,
                       uint64_t max_query_len, const char *seed_mask,
                       uint64_t num_partitions, uint64_t random_seed,
                       uint8_t *norm_text_out, uint64_t *sa_out, uint64_t *lcp_out, uint64_t cap,
                       uint64_t *num_suffixes_out, sufr_hip_stats *stats);

/* Exact LCP of two suffixes of a host text: the boundary fix of write(), find_lcp(prev.last,
 * this.first, text_len, 0) (sufr_builder.rs:893-902), used to stitch shards built on different GPUs. */
uint64_t sufr_hip_lcp_pair(const uint8_t *norm_text, uint64_t n, uint64_t a, uint64_t b);

/* ---- input and output formats ------------------------------------------------------------------
 * sufr_read_sequence_file replaces libsufr::util::read_sequence_file (util.rs:51-89): FASTA/FASTQ ->
 * text = s1 + delimiter + s2 + ... + '$', start offsets, names (header up to first whitespace).
 * Arrays are malloc'ed by the library; release them with sufr_sequence_data_free. */
typedef struct sufr_sequence_data {
    uint8_t  *seq;             /* SequenceFileData.seq */
    uint64_t  seq_len;
    uint64_t *start_positions; /* SequenceFileData.start_positions */
    char    **sequence_names;  /* SequenceFileData.sequence_names */
    uint64_t  num_sequences;
} sufr_sequence_data;
int  sufr_read_sequence_file(const char *path, uint8_t sequence_delimiter, sufr_sequence_data *out,
                             char *err, size_t errlen);
void sufr_sequence_data_free(sufr_sequence_data *d);

/* sufr_write_file replaces SufrBuilder::write (sufr_builder.rs:817-918): the version-6 .sufr layout,
 * byte for byte.  index_width is 4 or 8 (sizeof T); sequence_starts are given as u64 and stored
 * T-wide (857).  seed_mask NULL = none; has_max_query_len 0 = None. */
int sufr_write_file(const char *path, int is_dna, int allow_ambiguity, int ignore_softmask,
                    const uint8_t *norm_text, uint64_t text_len, int index_width,
                    const void *sa, const void *lcp, uint64_t num_suffixes,
                    int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                    const uint64_t *sequence_starts, uint64_t num_sequences,
                    const char *const *sequence_names, char *err, size_t errlen);

/* sufr_hip_create_file replaces sufr::create + SuffixArray::write (sufr/src/lib.rs:321-371,
 * suffix_array.rs:460-470): read the sequence file, build on the GPU, write `output`.
 * Arguments mirror CreateArgs (sufr/src/lib.rs:83-125). */
typedef struct sufr_create_args {
    const char *input;              /* <INPUT> */
    const char *output;             /* -o; NULL = "<input stem>.sufr" in the CWD */
    uint64_t    num_partitions;     /* -n, default 16 */
    int         has_max_query_len;  /* -m given */
    uint64_t    max_query_len;
    int         is_dna;             /* -d */
    int         allow_ambiguity;    /* -a */
    int         ignore_softmask;    /* -i */
    uint8_t     sequence_delimiter; /* -D, default '%' */
    const char *seed_mask;          /* -s */
    uint64_t    random_seed;        /* -r, default 42 */
} sufr_create_args;
int sufr_hip_create_file(sufr_hip_ctx *ctx, const sufr_create_args *args, char *path_out,
                         size_t path_out_len, sufr_hip_stats *stats);
/* The same from sequence data the caller has already read (sufr_read_sequence_file): lets a driver read
 * the file while the device context is being created.  `seq` is not modified and stays the caller's;
 * args->input only names the default output ("<input stem>.sufr"). */
int sufr_hip_create_from_sequence(sufr_hip_ctx *ctx, const sufr_sequence_data *seq,
                                  const sufr_create_args *args, char *path_out, size_t path_out_len,
                                  sufr_hip_stats *stats);


/* ---- several GPUs: shards by first-digit range, every shard written into its own range of the file ------------
 * replaces: the partition loop of SufrBuilder::write (sufr_builder.rs:875-906: SA and LCP of partition i appended
 * after those of the partitions before it, LCP[first of partition i > 0] := find_lcp(prev.last_suffix,
 * first_suffix, text_len, 0)) for shards built on different devices.
 *
 * One process driving N devices: sufr_hip_create_from_sequence_multi / sufr_hip_create_file_multi build shard r
 * of n_ctx on ctxs[r] (a host thread per context; contexts may share a device) and write ONE .sufr file, byte for
 * byte the file of the single-GPU build, --seed-mask and --max-query-len builds included (round 4; a cap below 8
 * symbols is built on ctxs[0] alone).  Texts that take windows (2^32 - 2^24 bytes and more) are sharded over the contexts
 * too since round 5 -- shard r of the windowed build on ctxs[r], arrays of the file's index width --, except seed-mask
 * builds, which ctxs[0] builds alone.  stats: n_ctx entries or NULL.
 *
 * One process per GPU (torch.distributed / MPI ranks): every rank calls sufr_hip_shard_build, the ranks exchange
 * their sufr_shard_info (24 bytes each: the only collective of the path), rank 0 calls sufr_write_frame, and after
 * a barrier every rank calls sufr_hip_shard_write with the suffix count of the ranks before it.
 * Texts that take windows too (round 5): the shard of the windowed build -- SA / LCP of the file's index width -- is kept
 * in device memory of its own, per context, until sufr_hip_shard_write has streamed it (or the context builds again or is
 * destroyed); seed-mask builds and caps below 8 symbols of such texts return SUFR_HIP_E_UNSUPPORTED for num_shards > 1.
 *
 * Device memory of one shard of N (text of n bytes, s suffixes in all): the text and its packed / bitmap forms
 * (~1.6 n), SA + LCP of the shard (8 s / N, sized from the shard's exact count), two record arrays (24 s / N) and the
 * level workspace -- the text is replicated, everything else scales with 1 / N. */
typedef struct sufr_shard_info {
    uint64_t num_suffixes;      /* suffixes of this shard */
    uint64_t first_suffix;      /* SA[0] of the shard (undefined when it is empty) */
    uint64_t last_suffix;       /* SA[num_suffixes - 1] */
} sufr_shard_info;
int sufr_hip_shard_build(sufr_hip_ctx *ctx, const sufr_sequence_data *seq, const sufr_create_args *args,
                         uint32_t shard_index, uint32_t num_shards, sufr_shard_info *info, sufr_hip_stats *stats);
/* creates / truncates `outfile` and writes the header and the name table of a file with total_suffixes suffixes */
int sufr_write_frame(const char *outfile, const sufr_sequence_data *seq, const sufr_create_args *args,
                     uint64_t total_suffixes, char *err, size_t errlen);
/* streams the resident shard (num_suffixes entries) to its place, suffix_offset entries into the SA / LCP sections
 * of the existing `outfile`; has_prev: LCP[0] of the shard is first set to the boundary LCP with prev_last_suffix;
 * write_text: this rank also writes the normalised text (rank 0).  CONSUMES the resident build: when it returns, the
 * context's SA / LCP / work arrays are released (and, with write_text == 0, its copy of the text); a second call without a
 * new sufr_hip_shard_build returns SUFR_HIP_E_INVALID. */
int sufr_hip_shard_write(sufr_hip_ctx *ctx, const sufr_sequence_data *seq, const sufr_create_args *args,
                         const char *outfile, uint64_t num_suffixes, uint64_t total_suffixes,
                         uint64_t suffix_offset, int has_prev, uint64_t prev_last_suffix, int write_text);
int sufr_hip_create_from_sequence_multi(sufr_hip_ctx *const *ctxs, int n_ctx, const sufr_sequence_data *seq,
                                        const sufr_create_args *args, char *path_out, size_t path_out_len,
                                        sufr_hip_stats *stats);
int sufr_hip_create_file_multi(sufr_hip_ctx *const *ctxs, int n_ctx, const sufr_create_args *args, char *path_out,
                               size_t path_out_len, sufr_hip_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* SUFR_HIP_H */
