// sufr_query.cpp -- the .sufr reader and the search of include/sufr_query.h (host C++, no device code).
//
// Replaces, for the query side of the reference: SufrFile::read (libsufr/src/sufr_file.rs:145-275),
// SufrSearch::search / suffix_search_first / suffix_search_last / compare (sufr_search.rs:104-350) and
// find_lcp_full_offset (util.rs:19-37).  The file is mapped, not read: text, SA and LCP are views into the mapping.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sufr_hip.h"
#include "../../include/sufr_query.h"

struct sufr_file {
    std::string path;
    const uint8_t* map = nullptr;
    size_t map_len = 0;
    sufr_file_meta meta{};
    const uint8_t* text = nullptr;
    const uint8_t* sa = nullptr;
    const uint8_t* lcp = nullptr;
    const uint8_t* mask = nullptr;            // seed_mask_len bytes of 0 / 1
    std::vector<uint64_t> seq_starts;
    std::vector<std::string> seq_names;
    std::vector<uint64_t> mask_positions;     // offsets of the 1s: the "care" positions (types.rs:36-200)
};

namespace {

void put_err(char* err, size_t errlen, const std::string& s)
{
    if (err && errlen) snprintf(err, errlen, "%s", s.c_str());
}

uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
uint64_t rdT(const uint8_t* p, int width, uint64_t i)
{
    if (width == 4) { uint32_t v; memcpy(&v, p + i * 4, 4); return v; }
    uint64_t v; memcpy(&v, p + i * 8, 8); return v;
}

struct Comparison { uint64_t lcp; int cmp; };          // cmp: -1 query < suffix, 0 equal, +1 query > suffix

// find_lcp_full_offset (util.rs:19-37): the text offset that follows `lcp` matched care positions
uint64_t full_offset(const sufr_file& f, uint64_t lcp)
{
    if (f.mask_positions.empty()) return lcp;
    if (lcp == 0 || lcp > f.meta.seed_mask_len) return lcp;
    // (lcp <= weight wherever compare() calls this: lcp counts care positions)
    const uint64_t offset = f.mask_positions[lcp - 1];
    const uint64_t next = lcp < f.mask_positions.size() ? f.mask_positions[lcp] : 0;
    return (next > offset && next - offset > 1) ? next : offset + 1;
}

// SufrSearch::compare (sufr_search.rs:241-343)
Comparison compare(const sufr_file& f, const uint8_t* q, size_t qlen, bool has_rt, uint64_t rt_mql, uint64_t suffix_pos,
                   uint64_t skip)
{
    const uint64_t n = f.meta.text_len;
    uint64_t lcp, max_query_len;
    if (f.mask_positions.empty()) {
        const uint64_t built = f.meta.max_query_len;
        max_query_len = (built > 0 && has_rt) ? (built < rt_mql ? built : rt_mql) : (has_rt ? rt_mql : built);
        if (max_query_len > 0 && skip >= max_query_len) lcp = skip;
        else {
            const uint64_t text_start = suffix_pos + skip;
            uint64_t text_end = max_query_len > 0 ? text_start + max_query_len : text_start + qlen;
            if (text_end > n) text_end = n;
            uint64_t k = 0;
            while (skip + k < qlen && text_start + k < text_end && q[skip + k] == f.text[text_start + k]) k++;
            lcp = skip + k;
        }
    } else {
        const uint64_t weight = f.mask_positions.size();
        max_query_len = has_rt ? rt_mql : 0;
        if (skip >= weight || (max_query_len > 0 && skip >= max_query_len)) lcp = skip;
        else {
            const uint64_t end = max_query_len > 0 ? (max_query_len < weight ? max_query_len : weight) : weight;
            uint64_t query_len = 0, suffix_len = 0;
            for (uint64_t i = skip; i < end; i++) {
                if (f.mask_positions[i] < qlen) query_len++;
                if (suffix_pos + f.mask_positions[i] < n) suffix_len++;
            }
            const uint64_t len = query_len < suffix_len ? query_len : suffix_len;
            uint64_t k = 0;
            while (k < len) {
                const uint64_t off = f.mask_positions[skip + k];
                if (suffix_pos + off >= n || q[off] != f.text[suffix_pos + off]) break;
                k++;
            }
            lcp = skip + k;
        }
    }
    int cmp;
    if (max_query_len > 0 && lcp >= max_query_len) cmp = 0;          // seen enough
    else {
        const uint64_t fo = full_offset(f, lcp);
        if (fo >= qlen) cmp = 0;                                      // the entire query matched
        else if (suffix_pos + fo >= n) cmp = 1;                       // (the reference has `unreachable!()` here)
        else cmp = q[fo] < f.text[suffix_pos + fo] ? -1 : (q[fo] > f.text[suffix_pos + fo] ? 1 : 0);
    }
    return {lcp, cmp};
}

}  // namespace

extern "C" {

int sufr_file_open(const char* path, sufr_file** out, char* err, size_t errlen)
{
    if (!path || !out) return SUFR_HIP_E_INVALID;
    *out = nullptr;
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { put_err(err, errlen, std::string(path) + ": " + strerror(errno)); return SUFR_HIP_E_IO; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 68) {
        put_err(err, errlen, std::string(path) + ": not a .sufr file (too short)");
        ::close(fd);
        return SUFR_HIP_E_IO;
    }
    void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { put_err(err, errlen, std::string(path) + ": mmap: " + strerror(errno)); return SUFR_HIP_E_IO; }
    sufr_file* f = new sufr_file;
    f->path = path; f->map = (const uint8_t*)m; f->map_len = (size_t)sb.st_size;
    auto fail = [&](const std::string& why) {
        put_err(err, errlen, std::string(path) + ": " + why);
        sufr_file_close(f);
        return SUFR_HIP_E_IO;
    };
    const uint8_t* p = f->map;
    sufr_file_meta& M = f->meta;
    M.version = p[0]; M.is_dna = p[1] == 1; M.allow_ambiguity = p[2] == 1; M.ignore_softmask = p[3] == 1;
    M.text_len = rd64(p + 4); M.text_pos = rd64(p + 12); M.suffix_array_pos = rd64(p + 20); M.lcp_pos = rd64(p + 28);
    M.len_suffixes = rd64(p + 36); M.max_query_len = rd64(p + 44); M.num_sequences = rd64(p + 52);
    M.index_width = M.text_len < 0xFFFFFFFFull ? 4 : 8;                // suffix_array.rs: u32 iff text_len < u32::MAX
    M.file_size = (uint64_t)sb.st_size; M.modified = (int64_t)sb.st_mtime;
    if (M.version != 6) return fail("unsupported .sufr version " + std::to_string(M.version) + " (this reader takes version 6)");
    const uint64_t W = (uint64_t)M.index_width;
    uint64_t at = 60;
    if (M.num_sequences > (f->map_len - at) / W) return fail("corrupt header (sequence starts)");
    f->seq_starts.resize(M.num_sequences);
    for (uint64_t i = 0; i < M.num_sequences; i++) f->seq_starts[i] = rdT(p + at, M.index_width, i);
    at += M.num_sequences * W;
    if (at + 8 > f->map_len) return fail("corrupt header (seed mask)");
    M.seed_mask_len = rd64(p + at); at += 8;
    if (M.seed_mask_len > f->map_len - at) return fail("corrupt header (seed mask)");
    if (M.seed_mask_len) {
        f->mask = p + at;
        for (uint64_t i = 0; i < M.seed_mask_len; i++) if (f->mask[i] == 1) f->mask_positions.push_back(i);
        at += M.seed_mask_len;
    }
    const uint64_t L = f->map_len;                                      // every bound without an overflowing product
    if (M.text_pos != at || M.text_len > L - at || M.len_suffixes > L / W || M.suffix_array_pos > L ||
        M.len_suffixes * W > L - M.suffix_array_pos || M.lcp_pos > L || M.len_suffixes * W > L - M.lcp_pos)
        return fail("corrupt header (section offsets)");
    f->text = p + M.text_pos; f->sa = p + M.suffix_array_pos; f->lcp = p + M.lcp_pos;
    // sequence names: bincode 1.x Vec<String> after the LCP section (u64 count, then u64 length + bytes each)
    uint64_t q = M.lcp_pos + M.len_suffixes * W;
    if (q + 8 > f->map_len) return fail("corrupt file (sequence names)");
    const uint64_t cnt = rd64(p + q); q += 8;
    for (uint64_t i = 0; i < cnt; i++) {
        if (q + 8 > f->map_len) return fail("corrupt file (sequence names)");
        const uint64_t len = rd64(p + q); q += 8;
        if (len > f->map_len - q) return fail("corrupt file (sequence names)");
        f->seq_names.emplace_back((const char*)p + q, (size_t)len);
        q += len;
    }
    if (f->seq_names.size() != M.num_sequences) return fail("corrupt file (sequence names do not match the header)");
    *out = f;
    return 0;
}

void sufr_file_close(sufr_file* f)
{
    if (!f) return;
    if (f->map) munmap((void*)f->map, f->map_len);
    delete f;
}

int sufr_file_metadata(const sufr_file* f, sufr_file_meta* meta)
{
    if (!f || !meta) return SUFR_HIP_E_INVALID;
    *meta = f->meta;
    return 0;
}

const uint8_t* sufr_file_text(const sufr_file* f) { return f ? f->text : nullptr; }
const uint8_t* sufr_file_seed_mask(const sufr_file* f) { return f ? f->mask : nullptr; }
const void* sufr_file_suffix_array(const sufr_file* f) { return f ? f->sa : nullptr; }
const void* sufr_file_lcp_array(const sufr_file* f) { return f ? f->lcp : nullptr; }
uint64_t sufr_file_suffix(const sufr_file* f, uint64_t rank) { return rdT(f->sa, f->meta.index_width, rank); }
uint64_t sufr_file_lcp(const sufr_file* f, uint64_t rank) { return rdT(f->lcp, f->meta.index_width, rank); }
uint64_t sufr_file_sequence_start(const sufr_file* f, uint64_t i) { return f->seq_starts[i]; }
const char* sufr_file_sequence_name(const sufr_file* f, uint64_t i) { return f->seq_names[i].c_str(); }

uint64_t sufr_file_sequence_of(const sufr_file* f, uint64_t pos)
{
    uint64_t lo = 0, hi = f->seq_starts.size();            // partition_point(|v| v <= pos)
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (f->seq_starts[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return lo ? lo - 1 : 0;
}

int sufr_file_search(const sufr_file* f, const uint8_t* q, size_t qlen, int has_mql, uint64_t mql, uint64_t* rank_lo,
                     uint64_t* rank_hi)
{
    if (!f || (!q && qlen) || !rank_lo || !rank_hi) return 0;
    const uint64_t n = f->meta.len_suffixes;
    if (n == 0) return 0;
    const bool rt = has_mql != 0;
    // suffix_search_first (sufr_search.rs:171-203), iteratively
    uint64_t first = 0;
    bool found = false;
    {
        uint64_t low = 0, high = n - 1, left = 0, right = 0;
        for (;;) {
            if (high < low) break;
            const uint64_t mid = low + (high - low) / 2;
            const uint64_t mv = sufr_file_suffix(f, mid);
            const Comparison c = compare(*f, q, qlen, rt, mql, mv, left < right ? left : right);
            if (c.cmp == 0 && (mid == 0 || compare(*f, q, qlen, rt, mql, sufr_file_suffix(f, mid - 1), 0).cmp > 0)) {
                first = mid; found = true; break;
            }
            if (c.cmp > 0) { low = mid + 1; left = c.lcp; }
            else { if (mid == 0) break; high = mid - 1; right = c.lcp; }
        }
    }
    if (!found) return 0;
    // suffix_search_last (206-238)
    uint64_t last = first;
    {
        uint64_t low = first, high = n - 1, left = 0, right = 0;
        for (;;) {
            if (high < low) break;
            const uint64_t mid = low + (high - low) / 2;
            const uint64_t mv = sufr_file_suffix(f, mid);
            const Comparison c = compare(*f, q, qlen, rt, mql, mv, left < right ? left : right);
            if (c.cmp == 0 && (mid == n - 1 || compare(*f, q, qlen, rt, mql, sufr_file_suffix(f, mid + 1), 0).cmp < 0)) {
                last = mid; break;
            }
            if (c.cmp < 0) { if (mid == 0) break; high = mid - 1; right = c.lcp; }
            else { low = mid + 1; left = c.lcp; }
        }
    }
    *rank_lo = first; *rank_hi = last + 1;
    return 1;
}

int sufr_file_search_batch(const sufr_file* f, const uint8_t* queries, const uint64_t* offsets, uint64_t nq, int has_mql,
                           uint64_t mql, uint64_t* rank_lo, uint64_t* rank_hi, int threads)
{
    if (!f || (nq && (!offsets || !rank_lo || !rank_hi))) return SUFR_HIP_E_INVALID;
    unsigned T = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (T == 0) T = 1;
    if (T > nq / 64 + 1) T = (unsigned)(nq / 64 + 1);          // a worker is not worth fewer than 64 queries
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const uint64_t b = next.fetch_add(256);
            if (b >= nq) return;
            const uint64_t e = b + 256 < nq ? b + 256 : nq;
            for (uint64_t i = b; i < e; i++) {
                uint64_t lo = 0, hi = 0;
                const int hit = sufr_file_search(f, queries + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), has_mql, mql, &lo, &hi);
                rank_lo[i] = hit ? lo : 0; rank_hi[i] = hit ? hi : 0;
            }
        }
    };
    if (T == 1) { worker(); return 0; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    return 0;
}

}  // extern "C"
