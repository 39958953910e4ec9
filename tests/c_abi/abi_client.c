/* A plain C client of the two public headers: what a host in another language binds.  Built by
 * tests/test_c_abi_client.py with gcc against libsufr_hip.so; no Python, no torch in this process.
 *
 *   abi_client query  <file.sufr> <query>...      host reader + search: "<query> <lo> <hi>" per query, then "names: ..."
 *   abi_client device <file.sufr> <query>...      the same ranges from the batched device search + positions of the first query
 *   abi_client build  <text-file> <out.sufr>      sufr_hip_build_u32 on the bytes of a file (--dna), written with sufr_write_file
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sufr_hip.h"
#include "sufr_query.h"

static int fail(const char *what, const char *msg) { fprintf(stderr, "%s: %s\n", what, msg ? msg : "?"); return 1; }

int main(int argc, char **argv)
{
    if (argc < 4) return fail("usage", "abi_client query|device|build ...");
    char err[512] = {0};
    if (!strcmp(argv[1], "query") || !strcmp(argv[1], "device")) {
        sufr_file *f = NULL;
        if (sufr_file_open(argv[2], &f, err, sizeof err) != 0) return fail("open", err);
        sufr_file_meta m;
        sufr_file_metadata(f, &m);
        const int nq = argc - 3;
        uint64_t *lo = calloc(nq, 8), *hi = calloc(nq, 8), *off = calloc(nq + 1, 8);
        size_t total = 0;
        for (int i = 0; i < nq; i++) total += strlen(argv[3 + i]);
        uint8_t *bytes = malloc(total + 1);
        for (int i = 0; i < nq; i++) { memcpy(bytes + off[i], argv[3 + i], strlen(argv[3 + i])); off[i + 1] = off[i] + strlen(argv[3 + i]); }
        if (!strcmp(argv[1], "query")) {
            if (sufr_file_search_batch(f, bytes, off, nq, 0, 0, lo, hi, 2) != 0) return fail("search", "batch failed");
        } else {
            sufr_hip_ctx *ctx = sufr_hip_create(0);
            if (!ctx) return fail("create", sufr_hip_last_error(NULL));
            sufr_hip_index *ix = NULL;
            if (sufr_hip_index_load(ctx, f, &ix) != 0) return fail("index", sufr_hip_last_error(ctx));
            if (sufr_hip_search_batch(ctx, ix, bytes, off, nq, 0, 0, lo, hi) != 0) return fail("search", sufr_hip_last_error(ctx));
            sufr_hip_index_free(ix);
            sufr_hip_destroy(ctx);
        }
        for (int i = 0; i < nq; i++) printf("%s %llu %llu\n", argv[3 + i], (unsigned long long)lo[i], (unsigned long long)hi[i]);
        printf("positions:");
        for (uint64_t r = lo[0]; r < hi[0]; r++) printf(" %llu", (unsigned long long)sufr_file_suffix(f, r));
        printf("\nnames:");
        for (uint64_t i = 0; i < m.num_sequences; i++) printf(" %s@%llu", sufr_file_sequence_name(f, i), (unsigned long long)sufr_file_sequence_start(f, i));
        printf("\n");
        sufr_file_close(f);
        return 0;
    }
    if (!strcmp(argv[1], "build")) {
        FILE *in = fopen(argv[2], "rb");
        if (!in) return fail("open", argv[2]);
        fseek(in, 0, SEEK_END);
        const long n = ftell(in);
        fseek(in, 0, SEEK_SET);
        uint8_t *text = malloc(n), *norm = malloc(n);
        if (fread(text, 1, n, in) != (size_t)n) return fail("read", argv[2]);
        fclose(in);
        uint32_t *sa = malloc(4 * n), *lcp = malloc(4 * n);
        uint64_t s = 0;
        sufr_hip_stats st;
        sufr_hip_ctx *ctx = sufr_hip_create(0);
        if (!ctx) return fail("create", sufr_hip_last_error(NULL));
        if (sufr_hip_build_u32(ctx, text, n, SUFR_HIP_FLAG_DNA | SUFR_HIP_FLAG_RAW_TEXT, 0, NULL, 16, 42, norm, sa, lcp, n, &s, &st) != 0)
            return fail("build", sufr_hip_last_error(ctx));
        const uint64_t start = 0;
        const char *name = "1";
        if (sufr_write_file(argv[3], 1, 0, 0, norm, n, 4, sa, lcp, s, 0, 0, NULL, &start, 1, &name, err, sizeof err) != 0) return fail("write", err);
        printf("built %llu suffixes of %ld bytes in %.3f ms\n", (unsigned long long)s, n, st.ms_total);
        sufr_hip_destroy(ctx);
        return 0;
    }
    return fail("usage", "unknown mode");
}
