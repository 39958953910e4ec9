/*
 * sufr_oracle.h -- public entry points of the CPU oracle (TEST INFRASTRUCTURE
 * ONLY; see sufr_oracle.c).  Loaded through ctypes by tests/, smoke() and
 * bench.py's cpu_baseline leg.  Never included by the product.
 */
#ifndef SUFR_ORACLE_H
#define SUFR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t text_len, num_suffixes, num_pivots, num_over_partitions;
    double t_pivots, t_partition, t_sort, t_total; /* seconds; phases the reference logs
                                                      (sufr_builder.rs:428,476,589) */
} oracle_stats;

void sufr_oracle_normalize(const uint8_t *in, uint8_t *out, uint64_t n, int ignore_softmask);

/* SufrBuilder::<T>::new(args) minus file I/O: sort() + the SA/LCP stitch of
 * write().  width = sizeof(T) (4 or 8).  sa_out/lcp_out need room for n. */
int sufr_oracle_build(const uint8_t *norm_text, uint64_t n, int is_dna, int allow_ambiguity,
                      int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                      uint64_t num_partitions, uint64_t random_seed, int threads, int width,
                      void *sa_out, void *lcp_out, oracle_stats *st, char *err, size_t errlen);

int64_t sufr_oracle_kat(const char *what, const uint8_t *norm_text, uint64_t n,
                        int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                        uint64_t a, uint64_t b, uint64_t len, uint64_t skip,
                        const uint64_t *pivots, uint64_t num_pivots);

int sufr_oracle_write_file(const char *path, int is_dna, int allow_ambiguity, int ignore_softmask,
                           const uint8_t *norm_text, uint64_t text_len, int width,
                           const void *sa, const void *lcp, uint64_t num_suffixes,
                           int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                           const uint64_t *sequence_starts, uint64_t num_sequences,
                           const char *const *sequence_names, char *err, size_t errlen);

int sufr_oracle_read_sequence_file(const char *path, uint8_t delimiter, uint8_t **seq_out,
                                   uint64_t *seq_len, uint64_t **starts_out, char ***names_out,
                                   uint64_t *num_seqs, char *err, size_t errlen);

void sufr_oracle_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
