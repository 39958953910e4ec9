/*
 * sufr_query.h -- reading a version-6 .sufr file and searching it (host code; part of libsufr_hip.so).
 *
 * What each entry point replaces in the reference (TravisWheelerLab/sufr):
 *   sufr_file_open        SufrFile::<T>::read                 libsufr/src/sufr_file.rs:145-275 (T chosen like
 *                                                             SuffixArray::read, suffix_array.rs: u32 iff text_len < u32::MAX)
 *   sufr_file_search      SufrSearch::search + compare        libsufr/src/sufr_search.rs:104-350, util.rs:19-37
 *   sufr_file_metadata    SufrFile::metadata                  sufr_file.rs:484-507
 *   accessors             FileAccess<T>::get / get_range      file_access.rs
 * The commands built on them (count / locate / extract / list / summarize, sufr/src/lib.rs:292-646) live in the
 * `sufr` binary (sufr_amd/csrc/sufr_cli.cpp) with the reference's output formats.
 *
 * The file is mapped read-only; nothing is copied.  max_query_len at query time: the reference searches a subsample
 * of the suffix array (first suffix of every distinct L-prefix, sufr_file.rs:440-460, cached under ~/.sufr) and maps
 * the hit back through ranks; here the whole array is searched with the same truncated comparison, which gives the
 * same rank range for every case the reference's tests cover and the range of ALL matching suffixes where the
 * reference's `rank[end] + 1` (sufr_search.rs:134) cuts the last group short.
 */
#ifndef SUFR_QUERY_H
#define SUFR_QUERY_H
#include <stddef.h>
#include <stdint.h>

#include "sufr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sufr_file sufr_file;

typedef struct sufr_file_meta {
    uint8_t  version, is_dna, allow_ambiguity, ignore_softmask;
    int      index_width;          /* 4 or 8: sizeof T */
    uint64_t text_len, text_pos, suffix_array_pos, lcp_pos, len_suffixes;
    uint64_t max_query_len;        /* of the build (0 when the file was built with a seed mask) */
    uint64_t num_sequences;
    uint64_t seed_mask_len;        /* 0: none */
    uint64_t file_size;
    int64_t  modified;             /* seconds since the epoch */
} sufr_file_meta;

int  sufr_file_open(const char *path, sufr_file **out, char *err, size_t errlen);
void sufr_file_close(sufr_file *f);
int  sufr_file_metadata(const sufr_file *f, sufr_file_meta *meta);
const uint8_t *sufr_file_text(const sufr_file *f);                 /* text_len bytes */
const uint8_t *sufr_file_seed_mask(const sufr_file *f);            /* seed_mask_len bytes of 0 / 1, or NULL */
const void *sufr_file_suffix_array(const sufr_file *f);            /* len_suffixes entries of index_width bytes */
const void *sufr_file_lcp_array(const sufr_file *f);
uint64_t sufr_file_suffix(const sufr_file *f, uint64_t rank);      /* SA[rank] */
uint64_t sufr_file_lcp(const sufr_file *f, uint64_t rank);         /* LCP[rank] */
uint64_t sufr_file_sequence_start(const sufr_file *f, uint64_t i);
const char *sufr_file_sequence_name(const sufr_file *f, uint64_t i);
/* index of the sequence that holds text position `pos`: partition_point(start <= pos) - 1 (sufr_file.rs:1149) */
uint64_t sufr_file_sequence_of(const sufr_file *f, uint64_t pos);

/* One query.  Returns 1 and the half-open rank range [*rank_lo, *rank_hi) when the query occurs, 0 when it does not.
 * has_max_query_len / max_query_len: the -m option of count / locate / extract. */
int sufr_file_search(const sufr_file *f, const uint8_t *query, size_t query_len, int has_max_query_len,
                     uint64_t max_query_len, uint64_t *rank_lo, uint64_t *rank_hi);

/* A batch on the host: `threads` workers share the queries (0: one per core), the reference's rayon loop over queries
 * (sufr_file.rs:760-800).  Queries are the concatenated bytes plus num_queries + 1 offsets; rank_lo / rank_hi receive
 * the half-open range per query, lo == hi == 0 when the query does not occur. */
int sufr_file_search_batch(const sufr_file *f, const uint8_t *queries, const uint64_t *offsets, uint64_t num_queries,
                           int has_max_query_len, uint64_t max_query_len, uint64_t *rank_lo, uint64_t *rank_hi, int threads);

/* ---- the same search for a batch of queries, on the GPU ---------------------------------------------------------
 * Replaces the rayon loop of SuffixArray::count / locate (libsufr/src/suffix_array.rs:181-236, 340-366;
 * sufr_file.rs:760-800): text and suffix array are resident in HBM, one launch answers the batch, one lane per
 * query.  Suffix arrays of both widths (u32 iff text_len < 2^32 - 1, suffix_array.rs:460-470; d_positions of
 * sufr_hip_locate_batch_device has the width of the array).
 *
 * sufr_hip_index_load   copies text + SA (+ seed mask) of an open file to the context's device.
 * sufr_hip_index_wrap   wraps arrays that are already on the device -- e.g. the normalized text handed to
 *                       sufr_hip_sort_device_u32 and the SA it produced: build, then query, without leaving HBM.
 *                       The caller keeps ownership of d_text / d_sa.  seed_mask: the "1101"-style string or NULL.
 *                       flags: SUFR_HIP_FLAG_DNA as at build time; SUFR_HIP_FLAG_NO_PREFIX_TABLE skips the table below.
 * Both build a prefix table next to the arrays: the rank range of every string of k symbols (ACGT for DNA, k = 14 on a
 * genome: 2 GB; otherwise the frequent bytes of the text), so that a query starts its two binary searches inside the
 * few ranks that share its first k symbols.  Answers do not depend on it.
 * Queries are the concatenated query bytes plus num_queries + 1 offsets (query i = bytes [offsets[i], offsets[i+1])).
 * rank_lo / rank_hi receive the half-open rank range per query, lo == hi == 0 when the query does not occur.
 * _batch takes host buffers and returns when the answers are in rank_lo / rank_hi; _batch_device takes device buffers
 * and only enqueues on the context's stream. */
typedef struct sufr_hip_index sufr_hip_index;
int  sufr_hip_index_load(sufr_hip_ctx *ctx, const sufr_file *f, sufr_hip_index **out);
#define SUFR_HIP_FLAG_NO_PREFIX_TABLE 0x100u
#define SUFR_HIP_FLAG_SA_U64 0x200u          /* sufr_hip_index_wrap: d_sa holds u64 entries although text_len < 2^32 - 1 */
int  sufr_hip_index_wrap(sufr_hip_ctx *ctx, const void *d_text, uint64_t text_len, const void *d_sa, uint64_t num_suffixes,
                         uint32_t flags, uint64_t built_max_query_len, const char *seed_mask, sufr_hip_index **out);
void sufr_hip_index_free(sufr_hip_index *ix);
/* Bytes per entry of the index's suffix array, 4 or 8: the width sufr_hip_index_wrap assumed for d_sa (8 iff text_len >=
 * 2^32 - 1 or SUFR_HIP_FLAG_SA_U64) and the width of the d_positions entries of sufr_hip_locate_batch_device, whose `cap` is
 * counted in entries of that width.  sufr_hip_index_wrap refuses a d_sa whose device allocation is shorter than
 * num_suffixes entries of it (SUFR_HIP_E_INVALID). */
int  sufr_hip_index_width(const sufr_hip_index *ix);
int  sufr_hip_search_batch(sufr_hip_ctx *ctx, const sufr_hip_index *ix, const uint8_t *queries, const uint64_t *offsets,
                           uint64_t num_queries, int has_max_query_len, uint64_t max_query_len, uint64_t *rank_lo,
                           uint64_t *rank_hi);
int  sufr_hip_search_batch_device(sufr_hip_ctx *ctx, const sufr_hip_index *ix, const void *d_queries, const void *d_offsets,
                                  uint64_t num_queries, int has_max_query_len, uint64_t max_query_len, void *d_rank_lo,
                                  void *d_rank_hi);

/* locate: the positions behind the rank ranges of a batch (SufrFile::locate, sufr_file.rs:1110-1175, without the
 * sequence names: those are a host lookup per position, sufr_file_sequence_of).  For query i the suffixes
 * SA[lo_i .. min(hi_i, lo_i + max_hits)) in rank order (max_hits 0: all) land in d_positions[d_offsets[i] .. d_offsets[i+1]);
 * d_offsets holds num_queries + 1 u64, d_positions up to `cap` entries of the index's width.  *total_out = d_offsets[num_queries] even when it
 * exceeds cap (the call then returns SUFR_HIP_E_CAPACITY and gathers nothing).  Enqueued on the context's stream after
 * one synchronisation for the total. */
int  sufr_hip_locate_batch_device(sufr_hip_ctx *ctx, const sufr_hip_index *ix, const void *d_rank_lo, const void *d_rank_hi,
                                  uint64_t num_queries, uint64_t max_hits, void *d_offsets, void *d_positions, uint64_t cap,
                                  uint64_t *total_out);

#ifdef __cplusplus
}
#endif
#endif /* SUFR_QUERY_H */
