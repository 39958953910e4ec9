// sufr_io.cpp -- host-side formats of the construction path: sequence-file reader, text map,
// .sufr v6 writer, and the `sufr create` driver.  Plain C++ (no device code).
//
// Reference behaviour mirrored here:
//   read_sequence_file      libsufr/src/util.rs:51-89
//   text normalisation      libsufr/src/sufr_builder.rs:144-160
//   SufrBuilder::write      libsufr/src/sufr_builder.rs:817-918   (.sufr version 6)
//   sufr::create            sufr/src/lib.rs:321-371, width rule suffix_array.rs:460-470
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/sufr_hip.h"

namespace {

void put_err(char* err, size_t errlen, const std::string& s)
{
    if (err && errlen) snprintf(err, errlen, "%s", s.c_str());
}

// a read-only view of a whole file (mmap; empty files map to nothing)
struct FileView {
    const uint8_t* data = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const char* path, std::string& why)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { why = std::string(path) + ": " + strerror(errno); return false; }
        struct stat sb;
        if (fstat(fd, &sb) != 0) { why = std::string(path) + ": " + strerror(errno); return false; }
        size = (size_t)sb.st_size;
        if (size) {
            void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) { why = std::string(path) + ": mmap: " + strerror(errno); return false; }
            data = (const uint8_t*)p;
        }
        return true;
    }
    ~FileView()
    {
        if (data) munmap((void*)data, size);
        if (fd >= 0) close(fd);
    }
};

inline const uint8_t* find_byte(const uint8_t* p, const uint8_t* end, int c)
{
    const void* q = memchr(p, c, (size_t)(end - p));
    return q ? (const uint8_t*)q : end;
}

// append [p, q) to dst without '\r' (line terminators are not part of the sequence)
inline void append_line(std::vector<uint8_t>& dst, const uint8_t* p, const uint8_t* q)
{
    if (q > p && q[-1] == '\r') q--;
    const uint8_t* cr = (const uint8_t*)memchr(p, '\r', (size_t)(q - p));
    if (!cr) { dst.insert(dst.end(), p, q); return; }
    for (; p < q; p++) if (*p != '\r') dst.push_back(*p);
}

void le64(uint8_t* b, uint64_t v) { for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); }

struct Out {
    FILE* f = nullptr;
    bool bad = false;
    uint64_t pos = 0;
    void raw(const void* p, size_t nbytes)
    {
        if (nbytes && fwrite(p, 1, nbytes, f) != nbytes) bad = true;
        pos += nbytes;
    }
    void u64(uint64_t v) { uint8_t b[8]; le64(b, v); raw(b, 8); }   // usize_to_bytes, util.rs:138-152
};

}  // namespace

extern "C" void sufr_hip_set_error_(sufr_hip_ctx* ctx, const char* msg);  // sufr_capi.inc

extern "C" {

int sufr_hip_normalize(const uint8_t* in, uint8_t* out, uint64_t n, int ignore_softmask)
{
    if ((!in || !out) && n) return SUFR_HIP_E_INVALID;
    // lowercase ASCII -> 'N' when soft-masked regions are ignored, else to uppercase (b & 0b1011111)
    uint8_t map[256];
    for (int c = 0; c < 256; c++)
        map[c] = (c >= 97 && c <= 122) ? (uint8_t)(ignore_softmask ? 'N' : (c & 0x5F)) : (uint8_t)c;
    for (uint64_t i = 0; i < n; i++) out[i] = map[in[i]];
    return 0;
}

uint64_t sufr_hip_lcp_pair(const uint8_t* t, uint64_t n, uint64_t a, uint64_t b)
{
    if (a >= n || b >= n) return 0;
    uint64_t lim = n - (a > b ? a : b), k = 0;
    while (k + 8 <= lim) {
        uint64_t x, y;
        memcpy(&x, t + a + k, 8); memcpy(&y, t + b + k, 8);
        if (x != y) return k + (uint64_t)(__builtin_ctzll(x ^ y) >> 3);
        k += 8;
    }
    while (k < lim && t[a + k] == t[b + k]) k++;
    return k;
}

int sufr_read_sequence_file(const char* path, uint8_t delimiter, sufr_sequence_data* out, char* err,
                            size_t errlen)
{
    if (!path || !out) return SUFR_HIP_E_INVALID;
    memset(out, 0, sizeof *out);
    FileView fv;
    std::string why;
    if (!fv.open(path, why)) { put_err(err, errlen, why); return SUFR_HIP_E_IO; }
    const uint8_t* p = fv.data;
    const uint8_t* end = fv.data + fv.size;
    while (p < end && (*p == '\n' || *p == '\r' || *p == ' ' || *p == '\t')) p++;
    if (p >= end) { put_err(err, errlen, std::string(path) + ": empty sequence file"); return SUFR_HIP_E_IO; }
    if (fv.size >= 2 && fv.data[0] == 0x1f && fv.data[1] == 0x8b) {
        put_err(err, errlen, std::string(path) + ": compressed input is not supported");
        return SUFR_HIP_E_UNSUPPORTED;
    }
    if (*p != '>' && *p != '@') {
        put_err(err, errlen, std::string(path) + ": expected a FASTA ('>') or FASTQ ('@') record");
        return SUFR_HIP_E_IO;
    }
    const bool fastq = *p == '@';
    std::vector<uint8_t> seq;
    seq.reserve(fv.size + 1);
    std::vector<uint64_t> starts;
    std::vector<std::string> names;
    while (p < end) {
        // header: everything after the marker up to the end of line
        const uint8_t* hb = p + 1;
        const uint8_t* he = find_byte(hb, end, '\n');
        p = he < end ? he + 1 : end;
        if (he > hb && he[-1] == '\r') he--;
        if (!starts.empty()) seq.push_back(delimiter);               // util.rs:62-64
        starts.push_back(seq.size());                                 // util.rs:67
        if (fastq) {
            const uint8_t* le = find_byte(p, end, '\n');
            append_line(seq, p, le);
            p = le < end ? le + 1 : end;
            for (int skip = 0; skip < 2; skip++) {                   // '+' line, quality line
                const uint8_t* e2 = find_byte(p, end, '\n');
                p = e2 < end ? e2 + 1 : end;
            }
            while (p < end && (*p == '\n' || *p == '\r')) p++;
        } else {
            while (p < end && *p != '>') {
                const uint8_t* le = find_byte(p, end, '\n');
                append_line(seq, p, le);
                p = le < end ? le + 1 : end;
            }
        }
        // id = header up to the first whitespace; the reference's fallback is (i+1) AFTER i was
        // incremented for this record (util.rs:70-77)
        const uint8_t* a = hb;
        while (a < he && (*a == ' ' || *a == '\t')) a++;
        const uint8_t* b = a;
        while (b < he && *b != ' ' && *b != '\t') b++;
        if (b > a) names.emplace_back((const char*)a, (size_t)(b - a));
        else names.push_back(std::to_string(starts.size() + 1));
    }
    seq.push_back(SUFR_SENTINEL_CHARACTER);                          // util.rs:83

    out->seq_len = seq.size();
    out->num_sequences = starts.size();
    out->seq = (uint8_t*)malloc(seq.size());
    out->start_positions = (uint64_t*)malloc(sizeof(uint64_t) * (starts.size() ? starts.size() : 1));
    out->sequence_names = (char**)calloc(starts.size() ? starts.size() : 1, sizeof(char*));
    if (!out->seq || !out->start_positions || !out->sequence_names) {
        sufr_sequence_data_free(out);
        put_err(err, errlen, "out of memory");
        return SUFR_HIP_E_NOMEM;
    }
    memcpy(out->seq, seq.data(), seq.size());
    for (size_t i = 0; i < starts.size(); i++) {
        out->start_positions[i] = starts[i];
        out->sequence_names[i] = strdup(names[i].c_str());
    }
    return 0;
}

void sufr_sequence_data_free(sufr_sequence_data* d)
{
    if (!d) return;
    free(d->seq);
    free(d->start_positions);
    if (d->sequence_names) {
        for (uint64_t i = 0; i < d->num_sequences; i++) free(d->sequence_names[i]);
        free(d->sequence_names);
    }
    memset(d, 0, sizeof *d);
}

int sufr_write_file(const char* path, int is_dna, int allow_ambiguity, int ignore_softmask,
                    const uint8_t* norm_text, uint64_t text_len, int index_width, const void* sa,
                    const void* lcp, uint64_t num_suffixes, int has_max_query_len, uint64_t max_query_len,
                    const char* seed_mask, const uint64_t* sequence_starts, uint64_t num_sequences,
                    const char* const* sequence_names, char* err, size_t errlen)
{
    if (!path || (index_width != 4 && index_width != 8)) return SUFR_HIP_E_INVALID;
    Out o;
    o.f = fopen(path, "wb");
    if (!o.f) { put_err(err, errlen, std::string(path) + ": " + strerror(errno)); return SUFR_HIP_E_IO; }
    static char iobuf[1 << 20];
    setvbuf(o.f, iobuf, _IOFBF, sizeof iobuf);
    const uint8_t head[4] = {SUFR_OUTFILE_VERSION, (uint8_t)(is_dna != 0), (uint8_t)(allow_ambiguity != 0),
                             (uint8_t)(ignore_softmask != 0)};
    o.raw(head, 4);
    o.u64(text_len);
    const uint64_t locs = o.pos;          // text_pos, sa_pos, lcp_pos: patched at the end
    o.u64(0); o.u64(0); o.u64(0);
    o.u64(num_suffixes);
    o.u64((!seed_mask && has_max_query_len) ? max_query_len : 0);
    o.u64(num_sequences);
    for (uint64_t i = 0; i < num_sequences; i++) {          // stored T-wide
        if (index_width == 4) { uint32_t v = (uint32_t)sequence_starts[i]; o.raw(&v, 4); }
        else { uint64_t v = sequence_starts[i]; o.raw(&v, 8); }
    }
    if (seed_mask) {
        size_t ml = strlen(seed_mask);
        o.u64(ml);
        std::vector<uint8_t> mb(ml);
        for (size_t i = 0; i < ml; i++) mb[i] = seed_mask[i] == '1';
        o.raw(mb.data(), ml);
    } else {
        o.u64(0);
    }
    const uint64_t text_pos = o.pos;
    o.raw(norm_text, text_len);
    const uint64_t sa_pos = o.pos;
    o.raw(sa, (size_t)(num_suffixes * (uint64_t)index_width));
    const uint64_t lcp_pos = o.pos;
    o.raw(lcp, (size_t)(num_suffixes * (uint64_t)index_width));
    // bincode 1.x Vec<String>: u64 count, then u64 length + UTF-8 bytes per name
    o.u64(num_sequences);
    for (uint64_t i = 0; i < num_sequences; i++) {
        size_t l = strlen(sequence_names[i]);
        o.u64(l);
        o.raw(sequence_names[i], l);
    }
    if (fseeko(o.f, (off_t)locs, SEEK_SET) != 0) o.bad = true;
    o.u64(text_pos); o.u64(sa_pos); o.u64(lcp_pos);
    if (fclose(o.f) != 0) o.bad = true;
    if (o.bad) { put_err(err, errlen, std::string(path) + ": write failed"); return SUFR_HIP_E_IO; }
    return 0;
}

int sufr_hip_create_file(sufr_hip_ctx* ctx, const sufr_create_args* a, char* path_out, size_t path_out_len,
                         sufr_hip_stats* stats)
{
    if (!ctx || !a || !a->input) return SUFR_HIP_E_INVALID;
    char err[512] = {0};
    sufr_sequence_data sd;
    int rc = sufr_read_sequence_file(a->input, a->sequence_delimiter ? a->sequence_delimiter : (uint8_t)'%', &sd,
                                     err, sizeof err);
    if (rc != 0) { sufr_hip_set_error_(ctx, err); return rc; }
    // default output name: "<input file stem>.sufr" in the current directory (sufr/src/lib.rs:334-340)
    std::string outfile;
    if (a->output) outfile = a->output;
    else {
        std::string in = a->input;
        size_t slash = in.find_last_of('/');
        std::string base = slash == std::string::npos ? in : in.substr(slash + 1);
        size_t dot = base.find_last_of('.');
        if (dot != std::string::npos && dot > 0) base = base.substr(0, dot);
        if (base.empty()) base = "out";
        outfile = base + ".sufr";
    }
    if (path_out && path_out_len) snprintf(path_out, path_out_len, "%s", outfile.c_str());
    const uint64_t n = sd.seq_len;
    const int width = n < 0xFFFFFFFFull ? 4 : 8;                     // suffix_array.rs:461
    uint32_t flags = SUFR_HIP_FLAG_RAW_TEXT;
    if (a->is_dna) flags |= SUFR_HIP_FLAG_DNA;
    if (a->allow_ambiguity) flags |= SUFR_HIP_FLAG_ALLOW_AMBIGUITY;
    if (a->ignore_softmask) flags |= SUFR_HIP_FLAG_IGNORE_SOFTMASK;
    std::vector<uint8_t> norm(n);
    uint64_t s = 0;
    void* sa = malloc((size_t)n * (size_t)width + 8);
    void* lcp = malloc((size_t)n * (size_t)width + 8);
    if (!sa || !lcp) { free(sa); free(lcp); sufr_sequence_data_free(&sd); return SUFR_HIP_E_NOMEM; }
    const uint64_t mql = a->has_max_query_len ? a->max_query_len : 0;
    if (a->has_max_query_len && a->seed_mask) {
        rc = SUFR_HIP_E_CONFLICT;    // clap's conflicts_with in the reference CLI; builder check 163-165
    } else if (width == 4) {
        rc = sufr_hip_build_u32(ctx, sd.seq, n, flags, mql, a->seed_mask, a->num_partitions, a->random_seed,
                                norm.data(), (uint32_t*)sa, (uint32_t*)lcp, n, &s, stats);
    } else {
        rc = sufr_hip_build_u64(ctx, sd.seq, n, flags, mql, a->seed_mask, a->num_partitions, a->random_seed,
                                norm.data(), (uint64_t*)sa, (uint64_t*)lcp, n, &s, stats);
    }
    if (rc == 0) {
        rc = sufr_write_file(outfile.c_str(), a->is_dna, a->allow_ambiguity, a->ignore_softmask, norm.data(), n,
                             width, sa, lcp, s, a->has_max_query_len, a->max_query_len, a->seed_mask,
                             sd.start_positions, sd.num_sequences, (const char* const*)sd.sequence_names, err,
                             sizeof err);
        if (rc != 0) sufr_hip_set_error_(ctx, err);
    } else if (rc == SUFR_HIP_E_CONFLICT) {
        sufr_hip_set_error_(ctx, "Cannot use max_query_len and seed_mask together");
    }
    free(sa); free(lcp);
    sufr_sequence_data_free(&sd);
    return rc;
}

}  // extern "C"
