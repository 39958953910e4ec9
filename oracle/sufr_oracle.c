/*
 * sufr_oracle.c -- CPU oracle for the suffix-array + LCP construction path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (TravisWheelerLab/sufr v0.7.12, libsufr::sufr_builder: random
 * pivots -> upper_bound partitioning -> per-partition LCP merge sort ->
 * boundary-LCP stitch -> .sufr v6 serialisation).  It is the checker that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg compare
 * the HIP path against, and the timed CPU baseline ("kind": "port").  The
 * product (sufr_amd/, libsufr_hip.so) never links, loads or calls it.
 *
 * Parity pin: the reference is Rust and cannot be compiled in this image
 * (no cargo/rustc; crates not vendored), so the oracle is pinned against the
 * reference's own golden vectors instead: all 14 current-format
 * data/expected/ .sufr files (whole-file byte equality, SA *and* LCP) and the
 * inline known-answer tests of libsufr/src/lib.rs:45-365 and
 * libsufr/src/sufr_builder.rs:1042-1405 -- see tests/test_oracle_golden.py.
 * Unpinned corner: builds with --max-query-len > 0 (no golden build exists in
 * the reference; only the is_less KAT sufr_builder.rs:1084-1126 covers it).
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <errno.h>

#include "sufr_oracle.h"

typedef struct { volatile uint64_t v; } oracle_atomic_u64;
static inline void oracle_atomic_init(oracle_atomic_u64 *a) { a->v = 0; }
static inline uint64_t oracle_fetch_add(oracle_atomic_u64 *a, uint64_t d)
{
    return __atomic_fetch_add(&a->v, d, __ATOMIC_RELAXED);
}

static double oracle_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline uint64_t oracle_splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Builder state: the fields of SufrBuilder<T> that the comparison functions
 * read (sufr_builder.rs:38-89). */
typedef struct {
    const uint8_t *text;      /* normalised text */
    uint64_t text_len;
    int is_dna, allow_ambiguity;
    uint64_t max_query_len;   /* SuffixSortType::MaxQueryLen(v); 0 = full sort */
    int has_mask;             /* SuffixSortType::Mask */
    uint8_t *mask_bytes; uint64_t mask_len;
    uint64_t *mask_positions; uint64_t mask_weight;
    uint64_t *n_starts, *n_ends; uint64_t num_n_ranges; /* n_ranges */
} oracle_ctx;

/* find_n_run: sufr_builder.rs:241-254 (binary search over sorted ranges) */
static inline int oracle_find_n_run(const oracle_ctx *cx, uint64_t suffix, uint64_t *end)
{
    uint64_t lo = 0, hi = cx->num_n_ranges;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (cx->n_starts[mid] <= suffix && suffix < cx->n_ends[mid]) { *end = cx->n_ends[mid]; return 1; }
        if (cx->n_starts[mid] < suffix) lo = mid + 1; else hi = mid;
    }
    return 0;
}

/* find_lcp_full_offset: util.rs:19-37 */
static inline uint64_t oracle_find_lcp_full_offset(const oracle_ctx *cx, uint64_t lcp)
{
    if (!cx->has_mask) return lcp;
    if (lcp == 0 || lcp > cx->mask_len) return lcp;
    uint64_t offset = cx->mask_positions[lcp - 1];
    uint64_t next_offset = lcp < cx->mask_weight ? cx->mask_positions[lcp] : 0;
    if (next_offset > offset && next_offset - offset > 1) return next_offset;
    return offset + 1;
}

#define IDX uint32_t
#define SFX(name) name##_u32
#include "sufr_oracle_body.inc"
#undef IDX
#undef SFX
#define IDX uint64_t
#define SFX(name) name##_u64
#include "sufr_oracle_body.inc"
#undef IDX
#undef SFX

/* SeedMask::is_valid, types.rs:163-166: regex ^1+0[01]*1$ */
static int oracle_mask_is_valid(const char *m)
{
    size_t n = strlen(m), i = 0;
    if (n < 3) return 0;
    while (i < n && m[i] == '1') i++;
    if (i == 0 || i >= n || m[i] != '0') return 0;
    for (; i < n; i++) if (m[i] != '0' && m[i] != '1') return 0;
    return m[n - 1] == '1';
}

/* text normalisation: sufr_builder.rs:144-160 */
void sufr_oracle_normalize(const uint8_t *in, uint8_t *out, uint64_t n, int ignore_softmask)
{
    for (uint64_t i = 0; i < n; i++) {
        uint8_t b = in[i];
        if (b >= 97 && b <= 122) out[i] = ignore_softmask ? (uint8_t)'N' : (uint8_t)(b & 0x5F);
        else out[i] = b;
    }
}

static void oracle_ctx_free(oracle_ctx *cx)
{
    free(cx->mask_bytes); free(cx->mask_positions); free(cx->n_starts); free(cx->n_ends);
}

/* SufrBuilder::new up to (not including) sort(): sufr_builder.rs:143-216 */
static int oracle_ctx_init(oracle_ctx *cx, const uint8_t *norm_text, uint64_t n, int is_dna,
                           int allow_ambiguity, uint64_t max_query_len, int has_mql,
                           const char *seed_mask, char *err, size_t errlen)
{
    memset(cx, 0, sizeof(*cx));
    cx->text = norm_text; cx->text_len = n; cx->is_dna = is_dna; cx->allow_ambiguity = allow_ambiguity;
    if (seed_mask && has_mql) {                                  /* 163-165 */
        snprintf(err, errlen, "Cannot use max_query_len and seed_mask together");
        return -1;
    }
    if (seed_mask) {                                             /* 167-169, types.rs:80-97 */
        if (!oracle_mask_is_valid(seed_mask)) {
            snprintf(err, errlen, "Invalid seed mask '%s'", seed_mask);
            return -1;
        }
        cx->has_mask = 1;
        cx->mask_len = strlen(seed_mask);
        cx->mask_bytes = (uint8_t *)malloc(cx->mask_len);
        cx->mask_positions = (uint64_t *)malloc(cx->mask_len * sizeof(uint64_t));
        for (uint64_t i = 0; i < cx->mask_len; i++) {
            cx->mask_bytes[i] = seed_mask[i] == '1';
            if (seed_mask[i] == '1') cx->mask_positions[cx->mask_weight++] = i;
        }
    } else {
        cx->max_query_len = has_mql ? max_query_len : 0;         /* 171 */
    }
    if (allow_ambiguity) {                                       /* 174-195 */
        uint64_t cap = 16, cnt = 0;
        cx->n_starts = (uint64_t *)malloc(cap * sizeof(uint64_t));
        cx->n_ends = (uint64_t *)malloc(cap * sizeof(uint64_t));
        int in_run = 0; uint64_t start = 0;
        for (uint64_t i = 0; i < n; i++) {
            if (norm_text[i] == 'N') {
                if (!in_run) { in_run = 1; start = i; }
            } else {
                if (in_run && i - start >= 1000) {
                    if (cnt == cap) {
                        cap *= 2;
                        cx->n_starts = (uint64_t *)realloc(cx->n_starts, cap * sizeof(uint64_t));
                        cx->n_ends = (uint64_t *)realloc(cx->n_ends, cap * sizeof(uint64_t));
                    }
                    cx->n_starts[cnt] = start; cx->n_ends[cnt] = i; cnt++;
                }
                in_run = 0;
            }
        }
        cx->num_n_ranges = cnt;
    }
    return 0;
}

int sufr_oracle_build(const uint8_t *norm_text, uint64_t n, int is_dna, int allow_ambiguity,
                      int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                      uint64_t num_partitions, uint64_t random_seed, int threads, int width,
                      void *sa_out, void *lcp_out, oracle_stats *st, char *err, size_t errlen)
{
    oracle_ctx cx;
    if (errlen) err[0] = 0;
    if (oracle_ctx_init(&cx, norm_text, n, is_dna, allow_ambiguity, max_query_len,
                        has_max_query_len, seed_mask, err, errlen) != 0) return -1;
    int rc;
    if (width == 4) rc = build_u32(&cx, num_partitions, random_seed, threads, (uint32_t *)sa_out,
                                   (uint32_t *)lcp_out, st, err, errlen);
    else if (width == 8) rc = build_u64(&cx, num_partitions, random_seed, threads, (uint64_t *)sa_out,
                                        (uint64_t *)lcp_out, st, err, errlen);
    else { snprintf(err, errlen, "width must be 4 or 8"); rc = -1; }
    oracle_ctx_free(&cx);
    return rc;
}

/* Known-answer-test hooks for sufr_builder.rs:1042-1405 */
int64_t sufr_oracle_kat(const char *what, const uint8_t *norm_text, uint64_t n,
                        int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                        uint64_t a, uint64_t b, uint64_t len, uint64_t skip,
                        const uint64_t *pivots, uint64_t num_pivots)
{
    oracle_ctx cx; char err[128];
    if (oracle_ctx_init(&cx, norm_text, n, 0, 0, max_query_len, has_max_query_len, seed_mask,
                        err, sizeof err) != 0) return -1;
    int64_t r = -2;
    if (!strcmp(what, "find_lcp")) r = (int64_t)find_lcp_u64(&cx, a, b, len, skip);
    else if (!strcmp(what, "is_less")) r = is_less_u64(&cx, a, b);
    else if (!strcmp(what, "upper_bound")) r = (int64_t)upper_bound_u64(&cx, a, pivots, num_pivots);
    else if (!strcmp(what, "full_offset")) r = (int64_t)oracle_find_lcp_full_offset(&cx, a);
    oracle_ctx_free(&cx);
    return r;
}

/* ---- .sufr v6 writer: sufr_builder.rs:817-918 -------------------------
 * usize_to_bytes = 8 little-endian bytes (util.rs:138-152); arrays are raw
 * native-endian T (util.rs:159-169); names are bincode 1.x Vec<String>
 * (u64 count, then u64 len + bytes per name).                              */
static int put_u64(FILE *f, uint64_t v)
{
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    return fwrite(b, 1, 8, f) == 8 ? 0 : -1;
}

int sufr_oracle_write_file(const char *path, int is_dna, int allow_ambiguity, int ignore_softmask,
                           const uint8_t *norm_text, uint64_t text_len, int width,
                           const void *sa, const void *lcp, uint64_t num_suffixes,
                           int has_max_query_len, uint64_t max_query_len, const char *seed_mask,
                           const uint64_t *sequence_starts, uint64_t num_sequences,
                           const char *const *sequence_names, char *err, size_t errlen)
{
    FILE *f = fopen(path, "wb");
    if (!f) { snprintf(err, errlen, "%s: %s", path, strerror(errno)); return -1; } /* 820 */
    uint8_t head[4] = {6, (uint8_t)!!is_dna, (uint8_t)!!allow_ambiguity, (uint8_t)!!ignore_softmask};
    uint64_t bytes_out = 0;
    int bad = 0;
    bad |= fwrite(head, 1, 4, f) != 4; bytes_out += 4;               /* 829-830 */
    bad |= put_u64(f, text_len); bytes_out += 8;                      /* 833 */
    long locs_pos = ftell(f);                                          /* 837 */
    bad |= put_u64(f, 0); bad |= put_u64(f, 0); bad |= put_u64(f, 0); bytes_out += 24;
    bad |= put_u64(f, num_suffixes); bytes_out += 8;                  /* 843 */
    bad |= put_u64(f, (seed_mask || !has_max_query_len) ? 0 : max_query_len); bytes_out += 8; /* 846-851 */
    bad |= put_u64(f, num_sequences); bytes_out += 8;                 /* 854 */
    for (uint64_t i = 0; i < num_sequences; i++) {                    /* 857: T-width */
        if (width == 4) { uint32_t v = (uint32_t)sequence_starts[i]; bad |= fwrite(&v, 4, 1, f) != 1; }
        else { uint64_t v = sequence_starts[i]; bad |= fwrite(&v, 8, 1, f) != 1; }
        bytes_out += (uint64_t)width;
    }
    if (seed_mask) {                                                  /* 860-867 */
        uint64_t ml = strlen(seed_mask);
        bad |= put_u64(f, ml); bytes_out += 8;
        for (uint64_t i = 0; i < ml; i++) { uint8_t b = seed_mask[i] == '1'; bad |= fwrite(&b, 1, 1, f) != 1; }
        bytes_out += ml;
    } else { bad |= put_u64(f, 0); bytes_out += 8; }
    uint64_t text_pos = bytes_out;                                    /* 870-872 */
    bad |= fwrite(norm_text, 1, text_len, f) != text_len; bytes_out += text_len;
    uint64_t sa_pos = bytes_out;                                      /* 875-881 */
    bad |= fwrite(sa, (size_t)width, num_suffixes, f) != num_suffixes; bytes_out += num_suffixes * (uint64_t)width;
    uint64_t lcp_pos = bytes_out;                                     /* 883-906 */
    bad |= fwrite(lcp, (size_t)width, num_suffixes, f) != num_suffixes;
    bad |= put_u64(f, num_sequences);                                 /* 909 */
    for (uint64_t i = 0; i < num_sequences; i++) {
        uint64_t l = strlen(sequence_names[i]);
        bad |= put_u64(f, l);
        bad |= fwrite(sequence_names[i], 1, l, f) != l;
    }
    bad |= fseek(f, locs_pos, SEEK_SET) != 0;                         /* 912-915 */
    bad |= put_u64(f, text_pos); bad |= put_u64(f, sa_pos); bad |= put_u64(f, lcp_pos);
    bad |= fclose(f) != 0;
    if (bad) { snprintf(err, errlen, "%s: write failed", path); return -1; }
    return 0;
}

/* ---- read_sequence_file: util.rs:51-89 --------------------------------
 * needletail 0.6 is a third-party crate that is not in the reference tree;
 * its documented FASTA/FASTQ record semantics are restated: FASTA sequence
 * lines are joined with line terminators removed; the record id is the
 * header line without its marker; the reference keeps the id up to the first
 * whitespace (util.rs:74-77).  Compressed input is not handled (unpinned).  */
int sufr_oracle_read_sequence_file(const char *path, uint8_t delimiter, uint8_t **seq_out,
                                   uint64_t *seq_len, uint64_t **starts_out, char ***names_out,
                                   uint64_t *num_seqs, char *err, size_t errlen)
{
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(err, errlen, "%s: %s", path, strerror(errno)); return -1; }
    fseek(f, 0, SEEK_END); long fsz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)fsz + 1);
    if (fsz > 0 && fread(buf, 1, (size_t)fsz, f) != (size_t)fsz) { fclose(f); free(buf); snprintf(err, errlen, "%s: read failed", path); return -1; }
    fclose(f);
    uint64_t n = (uint64_t)fsz, p = 0;
    while (p < n && (buf[p] == '\n' || buf[p] == '\r' || buf[p] == ' ' || buf[p] == '\t')) p++;
    if (p >= n) { free(buf); snprintf(err, errlen, "%s: empty or invalid sequence file", path); return -1; }
    if (buf[p] != '>' && buf[p] != '@') { free(buf); snprintf(err, errlen, "%s: not FASTA/FASTQ", path); return -1; }
    int fastq = buf[p] == '@';
    uint8_t *seq = (uint8_t *)malloc(n + 2);
    uint64_t sl = 0, cap = 16, cnt = 0;
    uint64_t *starts = (uint64_t *)malloc(cap * sizeof(uint64_t));
    char **names = (char **)malloc(cap * sizeof(char *));
    while (p < n) {
        /* header line */
        uint64_t hs = p + 1, he = hs;
        while (he < n && buf[he] != '\n') he++;
        uint64_t hend = he; if (hend > hs && buf[hend - 1] == '\r') hend--;
        p = he < n ? he + 1 : n;
        if (cnt > 0) seq[sl++] = delimiter;                       /* 62-64 */
        if (cnt == cap) { cap *= 2; starts = (uint64_t *)realloc(starts, cap * sizeof(uint64_t)); names = (char **)realloc(names, cap * sizeof(char *)); }
        starts[cnt] = sl;                                         /* 67 */
        if (fastq) {
            while (p < n && buf[p] != '\n') { if (buf[p] != '\r') seq[sl++] = buf[p]; p++; }
            if (p < n) p++;
            while (p < n && buf[p] != '\n') p++;                  /* '+' line */
            if (p < n) p++;
            while (p < n && buf[p] != '\n') p++;                  /* quality */
            if (p < n) p++;
        } else {
            while (p < n && buf[p] != '>') {
                while (p < n && buf[p] != '\n') { if (buf[p] != '\r') seq[sl++] = buf[p]; p++; }
                if (p < n) p++;
            }
        }
        /* id up to first whitespace; fallback (i+1) after the increment (70-77) */
        uint64_t a = hs; while (a < hend && (buf[a] == ' ' || buf[a] == '\t')) a++;
        uint64_t b = a; while (b < hend && buf[b] != ' ' && buf[b] != '\t') b++;
        char *name;
        if (b > a) { name = (char *)malloc(b - a + 1); memcpy(name, buf + a, b - a); name[b - a] = 0; }
        else { name = (char *)malloc(24); snprintf(name, 24, "%llu", (unsigned long long)(cnt + 2)); }
        names[cnt++] = name;
    }
    seq[sl++] = '$';                                              /* 83 */
    free(buf);
    *seq_out = seq; *seq_len = sl; *starts_out = starts; *names_out = names; *num_seqs = cnt;
    return 0;
}

void sufr_oracle_free(void *p) { free(p); }
