// sufr_io.cpp -- host-side formats of the construction path: sequence-file reader, text map,
// .sufr v6 writer, and the `sufr create` driver.  Plain C++ (no device code).
//
// Reference behaviour mirrored here:
//   read_sequence_file      libsufr/src/util.rs:51-89
//   text normalisation      libsufr/src/sufr_builder.rs:144-160
//   SufrBuilder::write      libsufr/src/sufr_builder.rs:817-918   (.sufr version 6)
//   sufr::create            sufr/src/lib.rs:321-371, width rule suffix_array.rs:460-470
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <zlib.h>

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

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

unsigned host_threads(unsigned cap)
{
    unsigned h = std::thread::hardware_concurrency();
    if (h == 0) h = 4;
    return h < cap ? h : cap;
}

// ---- multi-threaded FASTA reader (same result as the serial loop in sufr_read_sequence_file) ---------------
// The file is cut into line-aligned slices; a first sweep counts, per slice, the sequence bytes before its
// first header and after every header; prefix sums give every slice its place in the output; a second sweep
// copies the sequence bytes.  A line is a header iff it starts with '>' (util.rs:55-79 via needletail).
struct FaSlice {
    const uint8_t* b = nullptr;
    const uint8_t* e = nullptr;
    uint64_t lead = 0;                       // sequence bytes before the slice's first header
    std::vector<const uint8_t*> hdr;         // header lines (pointing at '>')
    std::vector<uint64_t> after;             // sequence bytes after each header, inside the slice
};

inline uint64_t line_payload(const uint8_t* p, const uint8_t* q)      // bytes of [p,q) that are not '\r'
{
    uint64_t c = (uint64_t)(q - p);
    while (p < q) {
        const uint8_t* r = (const uint8_t*)memchr(p, '\r', (size_t)(q - p));
        if (!r) break;
        c--; p = r + 1;
    }
    return c;
}

inline uint8_t* copy_payload(uint8_t* dst, const uint8_t* p, const uint8_t* q)
{
    while (p < q) {
        const uint8_t* r = (const uint8_t*)memchr(p, '\r', (size_t)(q - p));
        const uint8_t* stop = r ? r : q;
        memcpy(dst, p, (size_t)(stop - p));
        dst += stop - p;
        p = r ? r + 1 : q;
    }
    return dst;
}

void fasta_parallel(const uint8_t* p, const uint8_t* end, uint8_t delimiter, unsigned T, uint8_t* out,
                    uint64_t& out_len, std::vector<uint64_t>& starts, std::vector<std::string>& names)
{
    std::vector<FaSlice> sl(T);
    const uint64_t total = (uint64_t)(end - p);
    for (unsigned t = 0; t < T; t++) {
        const uint8_t* b = p + total * t / T;
        if (t > 0) {                                      // move to the start of the next line
            b = find_byte(b - 1, end, '\n');
            b = b < end ? b + 1 : end;
        }
        sl[t].b = b;
        if (t > 0) sl[t - 1].e = b;
    }
    sl[T - 1].e = end;
    auto sweep = [&](unsigned t) {
        FaSlice& s = sl[t];
        uint64_t* cur = &s.lead;
        for (const uint8_t* q = s.b; q < s.e;) {
            const uint8_t* le = find_byte(q, s.e, '\n');
            if (*q == '>') {
                s.hdr.push_back(q);
                s.after.push_back(0);
                cur = &s.after.back();
            } else {
                *cur += line_payload(q, le);
            }
            q = le < s.e ? le + 1 : s.e;
            if (!s.after.empty()) cur = &s.after.back();   // the vector may have moved
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(sweep, t);
        for (auto& x : th) x.join();
    }
    // place of every slice in the output, and the global index of its first header
    std::vector<uint64_t> at(T), first_rec(T);
    uint64_t pos = 0, rec = 0;
    for (unsigned t = 0; t < T; t++) {
        at[t] = pos; first_rec[t] = rec;
        pos += sl[t].lead;
        for (size_t h = 0; h < sl[t].hdr.size(); h++) {
            if (rec > 0) pos++;                            // delimiter before every record but the first
            pos += sl[t].after[h];
            rec++;
        }
    }
    out_len = pos;
    starts.assign(rec, 0);
    names.assign(rec, std::string());
    auto fill = [&](unsigned t) {
        const FaSlice& s = sl[t];
        uint8_t* dst = out + at[t];
        uint64_t r = first_rec[t];
        for (const uint8_t* q = s.b; q < s.e;) {
            const uint8_t* le = find_byte(q, s.e, '\n');
            if (*q == '>') {
                if (r > 0) *dst++ = delimiter;             // util.rs:62-64
                starts[r] = (uint64_t)(dst - out);         // util.rs:67
                const uint8_t* hb = q + 1;
                const uint8_t* he = le;
                if (he > hb && he[-1] == '\r') he--;
                const uint8_t* a = hb;
                while (a < he && (*a == ' ' || *a == '\t')) a++;
                const uint8_t* b = a;
                while (b < he && *b != ' ' && *b != '\t') b++;
                names[r] = b > a ? std::string((const char*)a, (size_t)(b - a)) : std::to_string(r + 2);
                r++;
            } else {
                dst = copy_payload(dst, q, le);
            }
            q = le < s.e ? le + 1 : s.e;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(fill, t);
    for (auto& x : th) x.join();
}

// ---- .sufr v6 header and tail (SufrBuilder::write, sufr_builder.rs:826-867, 909) -----------------------------
struct SufrLayout {
    std::vector<uint8_t> head;   // everything before the text, section offsets filled in
    std::vector<uint8_t> tail;   // bincode Vec<String> of the sequence names
    uint64_t text_pos = 0, sa_pos = 0, lcp_pos = 0, tail_pos = 0;
};

void push64(std::vector<uint8_t>& v, uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i))); }

SufrLayout sufr_layout(int is_dna, int allow_ambiguity, int ignore_softmask, uint64_t text_len, int index_width,
                       uint64_t num_suffixes, int has_max_query_len, uint64_t max_query_len, const char* seed_mask,
                       const uint64_t* sequence_starts, uint64_t num_sequences, const char* const* sequence_names)
{
    SufrLayout L;
    std::vector<uint8_t>& h = L.head;
    h.push_back(SUFR_OUTFILE_VERSION); h.push_back((uint8_t)(is_dna != 0));
    h.push_back((uint8_t)(allow_ambiguity != 0)); h.push_back((uint8_t)(ignore_softmask != 0));
    push64(h, text_len);
    const size_t locs = h.size();
    push64(h, 0); push64(h, 0); push64(h, 0);
    push64(h, num_suffixes);
    push64(h, (!seed_mask && has_max_query_len) ? max_query_len : 0);
    push64(h, num_sequences);
    for (uint64_t i = 0; i < num_sequences; i++) {          // stored T-wide (857)
        const uint64_t v = sequence_starts[i];
        for (int k = 0; k < index_width; k++) h.push_back((uint8_t)(v >> (8 * k)));
    }
    if (seed_mask) {
        const size_t ml = strlen(seed_mask);
        push64(h, ml);
        for (size_t i = 0; i < ml; i++) h.push_back(seed_mask[i] == '1');
    } else {
        push64(h, 0);
    }
    L.text_pos = h.size();
    L.sa_pos = L.text_pos + text_len;
    L.lcp_pos = L.sa_pos + num_suffixes * (uint64_t)index_width;
    L.tail_pos = L.lcp_pos + num_suffixes * (uint64_t)index_width;
    const uint64_t p3[3] = {L.text_pos, L.sa_pos, L.lcp_pos};
    for (int k = 0; k < 3; k++) le64(h.data() + locs + 8 * k, p3[k]);
    push64(L.tail, num_sequences);                          // bincode 1.x Vec<String>
    for (uint64_t i = 0; i < num_sequences; i++) {
        const size_t l = strlen(sequence_names[i]);
        push64(L.tail, l);
        L.tail.insert(L.tail.end(), sequence_names[i], sequence_names[i] + l);
    }
    return L;
}

bool pwrite_all(int fd, const void* buf, size_t len, uint64_t off)
{
    const uint8_t* p = (const uint8_t*)buf;
    while (len) {
        ssize_t w = pwrite(fd, p, len, (off_t)off);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; off += (uint64_t)w; len -= (size_t)w;
    }
    return true;
}

}  // namespace

extern "C" void sufr_hip_set_error_(sufr_hip_ctx* ctx, const char* msg);  // sufr_capi.inc
extern "C" int sufr_hip_is_wide_(const sufr_hip_ctx* ctx, uint64_t n);
extern "C" int sufr_hip_ctx_device_(const sufr_hip_ctx* ctx);
extern "C" uint64_t sufr_hip_array_budget_(const sufr_hip_ctx* ctx);
extern "C" int sufr_hip_build_resident_(sufr_hip_ctx* ctx, const uint8_t* text, uint64_t n, uint32_t flags,
                                        uint64_t max_query_len, const char* seed_mask, uint32_t shard_index,
                                        uint32_t num_shards, uint64_t* num_suffixes, sufr_hip_stats* stats,
                                        int* device, const void** d_text, const void** d_sa,
                                        const void** d_lcp);                               // sufr_capi.inc
extern "C" void sufr_hip_release_build_arrays_(sufr_hip_ctx* ctx, int also_text);
extern "C" void sufr_hip_release_wide_arrays_(sufr_hip_ctx* ctx);
extern "C" int sufr_hip_resident_ends_(sufr_hip_ctx* ctx, uint64_t s, uint64_t* first, uint64_t* last);
extern "C" int sufr_hip_resident_stitch_(sufr_hip_ctx* ctx, uint64_t n, uint64_t prev_last, uint64_t* lcp_out);
extern "C" int sufr_hip_resident_arrays_(sufr_hip_ctx* ctx, int* device, const void** d_text, const void** d_sa,
                                         const void** d_lcp);

namespace {

// default output name: "<input file stem>.sufr" in the current directory (sufr/src/lib.rs:334-340)
std::string output_name(const sufr_create_args* a)
{
    if (a->output) return a->output;
    std::string in = a->input;
    size_t slash = in.find_last_of('/');
    std::string base = slash == std::string::npos ? in : in.substr(slash + 1);
    size_t dot = base.find_last_of('.');
    if (dot != std::string::npos && dot > 0) base = base.substr(0, dot);
    if (base.empty()) base = "out";
    return base + ".sufr";
}

uint32_t build_flags(const sufr_create_args* a)
{
    uint32_t flags = SUFR_HIP_FLAG_RAW_TEXT;
    if (a->is_dna) flags |= SUFR_HIP_FLAG_DNA;
    if (a->allow_ambiguity) flags |= SUFR_HIP_FLAG_ALLOW_AMBIGUITY;
    if (a->ignore_softmask) flags |= SUFR_HIP_FLAG_IGNORE_SOFTMASK;
    return flags;
}

SufrLayout layout_of(const sufr_sequence_data& sd, const sufr_create_args* a, uint64_t num_suffixes)
{
    return sufr_layout(a->is_dna, a->allow_ambiguity, a->ignore_softmask, sd.seq_len, sd.seq_len < 0xFFFFFFFFull ? 4 : 8 /* suffix_array.rs:461 */, num_suffixes,
                       a->has_max_query_len, a->max_query_len, a->seed_mask, sd.start_positions, sd.num_sequences,
                       (const char* const*)sd.sequence_names);
}

// Device arrays -> their place in the file: a few threads, each copying 32 MB pieces into its own pinned buffer and
// pwrite()-ing them at their final offset.  On the test box the device-to-host side runs at ~37 GB/s and the page
// cache takes ~10 GB/s however many threads write (a shared mapping of the file instead of pwrite measured the same).
struct Section { const void* src; uint64_t bytes, file_off; };
int stream_sections(int device, int fd, const std::vector<Section>& secs)
{
    struct Piece { const uint8_t* src; uint64_t len, off; };
    std::vector<Piece> pieces;
    const uint64_t PIECE = (uint64_t)32 << 20;
    for (const Section& sc : secs)
        for (uint64_t o = 0; o < sc.bytes; o += PIECE)
            pieces.push_back({(const uint8_t*)sc.src + o, sc.bytes - o < PIECE ? sc.bytes - o : PIECE, sc.file_off + o});
    if (pieces.empty()) return 0;
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    // (buffered writes to ONE file are serialised by its inode lock: 9.9 GB/s on the test box with 1 or 8 writers, 71 GB/s into
    // 8 files -- profiles/r04_pagecache_write.txt; a few double-buffered workers keep that one writer fed)
    unsigned W = host_threads(4);
#ifdef SUFR_HIP_PROBES
    if (const char* e = getenv("SUFR_PROBE_WRITE_THREADS")) if (atoi(e) > 0) W = (unsigned)atoi(e);
#endif
    if (W > pieces.size()) W = (unsigned)pieces.size();
    // a worker keeps two pinned buffers: the copy of its next piece runs while it writes the one before (round 3 copied,
    // waited, wrote, one after the other: the page cache then saw ~10.6 of the 15 GB/s it takes from a hot buffer)
    auto worker = [&]() {
        void* pin[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&pin[0], PIECE, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&pin[1], PIECE, hipHostMallocDefault) != hipSuccess ||
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            failed = 1;
            for (void* q : pin) if (q) (void)hipHostFree(q);
            return;
        }
        size_t cur = next.fetch_add(1);
        int b = 0;
        if (cur < pieces.size() && hipMemcpyAsync(pin[0], pieces[cur].src, pieces[cur].len, hipMemcpyDeviceToHost, st) != hipSuccess) failed = 1;
        while (!failed && cur < pieces.size()) {
            if (hipStreamSynchronize(st) != hipSuccess) { failed = 1; break; }
            const size_t nx = next.fetch_add(1);
            if (nx < pieces.size() &&
                hipMemcpyAsync(pin[b ^ 1], pieces[nx].src, pieces[nx].len, hipMemcpyDeviceToHost, st) != hipSuccess) { failed = 1; break; }
            if (!pwrite_all(fd, pin[b], pieces[cur].len, pieces[cur].off)) { failed = 2; break; }
            cur = nx; b ^= 1;
        }
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
        for (void* q : pin) (void)hipHostFree(q);
    };
    std::vector<std::thread> th;
    for (unsigned w = 0; w < W; w++) th.emplace_back(worker);
    for (auto& x : th) x.join();
    return failed;        // 0 ok, 1 device-to-host copy failed, 2 write failed
}

}  // namespace

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

// ---------------------------------------------------------------------------------------------------------------
// bzip2 / xz input (needletail inflates both transparently in the reference, util.rs:55).  The image ships the run-time
// libraries without their headers, so the two stable C interfaces are declared here and resolved with dlopen; a box
// without the library gets a message that names it.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct BzStream {                      // bz_stream of bzlib.h (libbz2 1.0)
    char* next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
    char* next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
    void* state; void* (*bzalloc)(void*, int, int); void (*bzfree)(void*, void*); void* opaque;
};
struct LzmaStream {                    // lzma_stream of lzma/base.h (liblzma 5.x)
    const uint8_t* next_in; size_t avail_in; uint64_t total_in;
    uint8_t* next_out; size_t avail_out; uint64_t total_out;
    const void* allocator; void* internal; void* reserved_ptr[4]; uint64_t reserved_int1, reserved_int2;
    size_t reserved_int3, reserved_int4; int reserved_enum1, reserved_enum2;
};

// Known-answer streams: the two stream structs above are declared BY HAND, so a library whose layout differs (another major
// version, another ABI) would otherwise read garbage silently.  Before the first real input each library inflates 60 known
// bytes through exactly the code path below; anything but those bytes -- and the counters the structs claim to hold -- makes
// the reader refuse the format with a message.
const uint8_t KAT_PLAIN[61] = ">kat\nACGTNNNNacgt$%\n>kat\nACGTNNNNacgt$%\n>kat\nACGTNNNNacgt$%\n";
const uint8_t KAT_BZ2[71] = {66, 90, 104, 57, 49, 65, 89, 38, 83, 89, 113, 179, 145, 152, 0, 0, 8, 223, 128, 64, 16, 6, 0, 0, 1, 40, 129, 4, 0, 40, 136,
    4, 0, 32, 0, 49, 76, 0, 1, 31, 170, 132, 194, 104, 218, 105, 226, 97, 117, 225, 150, 152, 68, 109, 183, 169, 126, 227, 138, 165, 209, 119, 36,
    83, 133, 9, 7, 27, 57, 25, 128};
const uint8_t KAT_XZ[88] = {253, 55, 122, 88, 90, 0, 0, 4, 230, 214, 180, 70, 2, 0, 33, 1, 22, 0, 0, 0, 116, 47, 229, 163, 224, 0, 59, 0, 26, 93, 0, 31,
    26, 200, 39, 116, 88, 133, 123, 175, 94, 108, 220, 169, 185, 90, 174, 181, 95, 34, 109, 230, 135, 123, 58, 0, 0, 0, 0, 0, 101, 151, 250, 13,
    141, 141, 9, 53, 0, 1, 54, 60, 217, 88, 245, 134, 31, 182, 243, 125, 1, 0, 0, 0, 0, 4, 89, 90};

bool inflate_bzip2(const uint8_t* src, size_t len, std::vector<uint8_t>& out, std::string& why);
bool inflate_xz(const uint8_t* src, size_t len, std::vector<uint8_t>& out, std::string& why);
// 0: not tried, 1: passed, -1: failed (decided once per process; the test calls the inflater with `src` = the known stream)
std::atomic<int> g_bz_kat{0}, g_xz_kat{0};
bool kat_ok(std::atomic<int>& state, bool (*inflate)(const uint8_t*, size_t, std::vector<uint8_t>&, std::string&), const uint8_t* blob, size_t blen,
            const char* lib, std::string& why)
{
    int s = state.load();
    if (s == 0) {
        state.store(2);                                          // (the self-test itself runs the inflater: do not recurse)
        std::vector<uint8_t> got; std::string w;
        const bool ok = inflate(blob, blen, got, w) && got.size() == 60 && memcmp(got.data(), KAT_PLAIN, 60) == 0;
        state.store(s = ok ? 1 : -1);
    }
    if (s == -1) { why = std::string(lib) + " on this machine does not decode the built-in known-answer stream: its stream struct differs from "
                         "the one sufr_io.cpp declares -- this input format is refused rather than misread"; return false; }
    return true;
}

bool inflate_bzip2(const uint8_t* src, size_t len, std::vector<uint8_t>& out, std::string& why)
{
    void* h = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!h) h = dlopen("libbz2.so.1", RTLD_NOW);
    if (!h) { why = "bzip2 input needs libbz2.so.1.0, which this machine does not have"; return false; }
    auto init = (int (*)(BzStream*, int, int))dlsym(h, "BZ2_bzDecompressInit");
    auto run = (int (*)(BzStream*))dlsym(h, "BZ2_bzDecompress");
    auto fini = (int (*)(BzStream*))dlsym(h, "BZ2_bzDecompressEnd");
    if (!init || !run || !fini) { why = "libbz2 lacks the BZ2_bzDecompress interface"; dlclose(h); return false; }
    if (src != KAT_BZ2 && !kat_ok(g_bz_kat, inflate_bzip2, KAT_BZ2, sizeof KAT_BZ2, "libbz2", why)) { dlclose(h); return false; }
    out.resize(len * 5 + (1u << 20));
    size_t have = 0, at = 0;
    bool ok = true;
    while (ok && at < len) {                                     // concatenated streams, like bzcat
        BzStream st;
        memset(&st, 0, sizeof st);
        if (init(&st, 0, 0) != 0) { why = "BZ2_bzDecompressInit failed"; ok = false; break; }
        int rc = 0;
        while (rc == 0) {
            if (out.size() - have < (1u << 20)) out.resize(out.size() * 2);
            const size_t in_left = len - at, room = out.size() - have;
            st.next_in = (char*)src + at; st.avail_in = (unsigned)(in_left > (1u << 30) ? (1u << 30) : in_left);
            st.next_out = (char*)out.data() + have; st.avail_out = (unsigned)(room > (1u << 30) ? (1u << 30) : room);
            const unsigned in0 = st.avail_in, out0 = st.avail_out;
            rc = run(&st);
            at += in0 - st.avail_in; have += out0 - st.avail_out;
            if (rc == 0 && in0 == st.avail_in && out0 == st.avail_out) { rc = -7; }      // no progress: truncated input
        }
        if (src == KAT_BZ2 && (st.total_out_lo32 != 60u || st.total_in_lo32 != (unsigned)sizeof KAT_BZ2)) rc = -99;   // (the struct's counters)
        fini(&st);
        if (rc != 4) { why = "corrupt or truncated bzip2 stream"; ok = false; }             // BZ_STREAM_END
        while (ok && at < len && (src[at] == 0)) at++;                                      // padding between streams
        if (ok && at < len && !(len - at >= 3 && src[at] == 'B' && src[at + 1] == 'Z' && src[at + 2] == 'h')) break;   // trailing garbage: ignored
    }
    dlclose(h);
    out.resize(ok ? have : 0);
    return ok;
}

bool inflate_xz(const uint8_t* src, size_t len, std::vector<uint8_t>& out, std::string& why)
{
    void* h = dlopen("liblzma.so.5", RTLD_NOW);
    if (!h) { why = "xz input needs liblzma.so.5, which this machine does not have"; return false; }
    auto init = (int (*)(LzmaStream*, uint64_t, uint32_t))dlsym(h, "lzma_stream_decoder");
    auto run = (int (*)(LzmaStream*, int))dlsym(h, "lzma_code");
    auto fini = (void (*)(LzmaStream*))dlsym(h, "lzma_end");
    if (!init || !run || !fini) { why = "liblzma lacks the lzma_stream_decoder interface"; dlclose(h); return false; }
    if (src != KAT_XZ && !kat_ok(g_xz_kat, inflate_xz, KAT_XZ, sizeof KAT_XZ, "liblzma", why)) { dlclose(h); return false; }
    LzmaStream st;
    memset(&st, 0, sizeof st);
    if (init(&st, UINT64_MAX, 0x08u /* LZMA_CONCATENATED */) != 0) { why = "lzma_stream_decoder failed"; dlclose(h); return false; }
    out.resize(len * 5 + (1u << 20));
    st.next_in = src; st.avail_in = len;
    size_t have = 0;
    int rc = 0;
    while (rc == 0) {
        if (out.size() - have < (1u << 20)) out.resize(out.size() * 2);
        st.next_out = out.data() + have; st.avail_out = out.size() - have;
        const size_t out0 = st.avail_out;
        rc = run(&st, st.avail_in ? 0 /* LZMA_RUN */ : 3 /* LZMA_FINISH */);
        have += out0 - st.avail_out;
    }
    if (src == KAT_XZ && (st.total_out != 60u || st.total_in != sizeof KAT_XZ)) rc = -99;     // (the struct's counters)
    fini(&st);
    dlclose(h);
    if (rc != 1) { why = "corrupt or truncated xz stream"; out.clear(); return false; }      // LZMA_STREAM_END
    out.resize(have);
    return true;
}
}  // namespace

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
    // gzip input (needletail, which the reference reads with, decompresses transparently): inflated into
    // memory first (concatenated members included), then parsed like a plain file
    std::vector<uint8_t> inflated;
    if (fv.size >= 2 && fv.data[0] == 0x1f && fv.data[1] == 0x8b) {
        gzFile gz = gzopen(path, "rb");
        if (!gz) { put_err(err, errlen, std::string(path) + ": cannot open gzip stream"); return SUFR_HIP_E_IO; }
        gzbuffer(gz, 1u << 20);
        inflated.resize(fv.size * 4 + (1u << 20));
        size_t have = 0;
        for (;;) {
            if (inflated.size() - have < (1u << 20)) inflated.resize(inflated.size() * 2);
            const size_t room = inflated.size() - have;
            int got = gzread(gz, inflated.data() + have, (unsigned)(room > (1u << 30) ? (1u << 30) : room));
            if (got < 0) {
                int zerr = 0;
                const char* msg = gzerror(gz, &zerr);
                put_err(err, errlen, std::string(path) + ": " + (msg ? msg : "gzip error"));
                gzclose(gz);
                return SUFR_HIP_E_IO;
            }
            if (got == 0) break;
            have += (size_t)got;
        }
        gzclose(gz);
        inflated.resize(have);
        p = inflated.data();
        end = inflated.data() + inflated.size();
    } else if (fv.size >= 3 && fv.data[0] == 'B' && fv.data[1] == 'Z' && fv.data[2] == 'h') {
        if (!inflate_bzip2(fv.data, fv.size, inflated, why)) { put_err(err, errlen, std::string(path) + ": " + why); return SUFR_HIP_E_IO; }
        p = inflated.data(); end = inflated.data() + inflated.size();
    } else if (fv.size >= 6 && fv.data[0] == 0xfd && fv.data[1] == '7' && fv.data[2] == 'z' && fv.data[3] == 'X' && fv.data[4] == 'Z' &&
               fv.data[5] == 0) {
        if (!inflate_xz(fv.data, fv.size, inflated, why)) { put_err(err, errlen, std::string(path) + ": " + why); return SUFR_HIP_E_IO; }
        p = inflated.data(); end = inflated.data() + inflated.size();
    }
    while (p < end && (*p == '\n' || *p == '\r' || *p == ' ' || *p == '\t')) p++;
    if (p >= end) { put_err(err, errlen, std::string(path) + ": empty sequence file"); return SUFR_HIP_E_IO; }
    if (*p != '>' && *p != '@') {
        put_err(err, errlen, std::string(path) + ": expected a FASTA ('>') or FASTQ ('@') record");
        return SUFR_HIP_E_IO;
    }
    const bool fastq = *p == '@';
    const unsigned T = host_threads(32);
    if (!fastq && T > 1 && (size_t)(end - p) >= ((size_t)32 << 20) && !getenv("SUFR_SERIAL_READER")) {
        std::vector<uint64_t> pstarts;
        std::vector<std::string> pnames;
        // (2 MB-aligned and advised for transparent huge pages: 3 GB of text in 4 KB pages is 760 000 page faults while
        // the parser fills it and as many pages to hand back when the process ends -- 0.2 s of a 2.2 s `sufr create`)
        const size_t bytes = (((size_t)(end - p) + 2) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        uint8_t* buf = (uint8_t*)aligned_alloc((size_t)2 << 20, bytes);
        if (!buf) { put_err(err, errlen, "out of memory"); return SUFR_HIP_E_NOMEM; }
#ifdef MADV_HUGEPAGE
        (void)madvise(buf, bytes, MADV_HUGEPAGE);
#endif
        uint64_t len = 0;
        fasta_parallel(p, end, delimiter, T, buf, len, pstarts, pnames);
        buf[len++] = SUFR_SENTINEL_CHARACTER;                        // util.rs:83
        out->seq = buf;
        out->seq_len = len;
        out->num_sequences = pstarts.size();
        out->start_positions = (uint64_t*)malloc(sizeof(uint64_t) * (pstarts.size() ? pstarts.size() : 1));
        out->sequence_names = (char**)calloc(pstarts.size() ? pstarts.size() : 1, sizeof(char*));
        if (!out->start_positions || !out->sequence_names) {
            sufr_sequence_data_free(out);
            put_err(err, errlen, "out of memory");
            return SUFR_HIP_E_NOMEM;
        }
        for (size_t i = 0; i < pstarts.size(); i++) {
            out->start_positions[i] = pstarts[i];
            out->sequence_names[i] = strdup(pnames[i].c_str());
        }
        return 0;
    }
    std::vector<uint8_t> seq;
    seq.reserve((size_t)(end - p) + 1);
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
    const SufrLayout L = sufr_layout(is_dna, allow_ambiguity, ignore_softmask, text_len, index_width, num_suffixes,
                                     has_max_query_len, max_query_len, seed_mask, sequence_starts, num_sequences,
                                     sequence_names);
    Out o;
    o.f = fopen(path, "wb");
    if (!o.f) { put_err(err, errlen, std::string(path) + ": " + strerror(errno)); return SUFR_HIP_E_IO; }
    std::vector<char> iobuf(1 << 20);                  // per call: concurrent writers must not share a stdio buffer
    setvbuf(o.f, iobuf.data(), _IOFBF, iobuf.size());
    o.raw(L.head.data(), L.head.size());
    o.raw(norm_text, text_len);
    o.raw(sa, (size_t)(num_suffixes * (uint64_t)index_width));
    o.raw(lcp, (size_t)(num_suffixes * (uint64_t)index_width));
    o.raw(L.tail.data(), L.tail.size());
    if (fclose(o.f) != 0) o.bad = true;
    if (o.bad) {
        (void)unlink(path);                            // no valid header over short sections left behind
        put_err(err, errlen, std::string(path) + ": write failed");
        return SUFR_HIP_E_IO;
    }
    return 0;
}

int sufr_write_frame(const char* outfile, const sufr_sequence_data* sd, const sufr_create_args* a,
                     uint64_t total_suffixes, char* err, size_t errlen)
{
    if (!outfile || !sd || !a) return SUFR_HIP_E_INVALID;
    const SufrLayout L = layout_of(*sd, a, total_suffixes);
    int fd = ::open(outfile, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { put_err(err, errlen, std::string(outfile) + ": " + strerror(errno)); return SUFR_HIP_E_IO; }
    bool ok = pwrite_all(fd, L.head.data(), L.head.size(), 0) && pwrite_all(fd, L.tail.data(), L.tail.size(), L.tail_pos);
    if (close(fd) != 0) ok = false;
    if (!ok) { (void)unlink(outfile); put_err(err, errlen, std::string(outfile) + ": write failed"); return SUFR_HIP_E_IO; }
    return 0;
}

// ---- the one-rank-per-GPU writer for texts that take windows (round 5) -----------------------------------------------
// sufr_hip_shard_build keeps such a shard outside the context's own build arrays: the raw text, and SA / LCP of the file's
// index width, in device memory of their own, remembered per context until sufr_hip_shard_write has streamed them (or the
// context builds again / is destroyed: sufr_io_release_ctx_, called by sufr_hip_destroy).
namespace {
struct WideResident { void *d_text = nullptr, *d_sa = nullptr, *d_lcp = nullptr; uint64_t s = 0, n = 0, first = 0; int width = 0, device = 0; };
std::mutex g_wide_mu;
std::unordered_map<const sufr_hip_ctx*, WideResident> g_wide;
void wide_free(WideResident& w)
{
    if (hipSetDevice(w.device) == hipSuccess)
        for (void* q : {w.d_text, w.d_sa, w.d_lcp}) if (q) (void)hipFree(q);
    w = WideResident();
}
}  // namespace

extern "C" void sufr_io_release_ctx_(const sufr_hip_ctx* ctx)
{
    WideResident w;
    {
        std::lock_guard<std::mutex> g(g_wide_mu);
        auto it = g_wide.find(ctx);
        if (it == g_wide.end()) return;
        w = it->second;
        g_wide.erase(it);
    }
    wide_free(w);
}

static int wide_shard_build(sufr_hip_ctx* ctx, const sufr_sequence_data* sd, const sufr_create_args* a, uint32_t shard_index,
                            uint32_t num_shards, sufr_shard_info* info, sufr_hip_stats* stats)
{
    sufr_io_release_ctx_(ctx);
    const uint64_t n = sd->seq_len;
    WideResident w;
    w.n = n; w.width = n < 0xFFFFFFFFull ? 4 : 8; w.device = sufr_hip_ctx_device_(ctx);
    const uint32_t flags = build_flags(a);
    const uint64_t mql = a->has_max_query_len ? a->max_query_len : 0;
    if (hipSetDevice(w.device) != hipSuccess || hipMalloc(&w.d_text, n + 64) != hipSuccess ||
        hipMemcpy(w.d_text, sd->seq, n, hipMemcpyHostToDevice) != hipSuccess) {
        wide_free(w);
        sufr_hip_set_error_(ctx, "out of device memory (text of a windowed shard)");
        return SUFR_HIP_E_NOMEM;
    }
    uint64_t cap = n / num_shards + n / (4ull * num_shards) + ((uint64_t)1 << 20);
    int rc = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (cap > n) cap = n;
        if (hipMalloc(&w.d_sa, cap * (size_t)w.width + 16) != hipSuccess || hipMalloc(&w.d_lcp, cap * (size_t)w.width + 16) != hipSuccess) {
            wide_free(w);
            sufr_hip_set_error_(ctx, "out of device memory (arrays of a windowed shard)");
            return SUFR_HIP_E_NOMEM;
        }
        w.s = 0;
        rc = w.width == 4
            ? sufr_hip_sort_device_u32(ctx, w.d_text, n, flags, mql, a->seed_mask, a->num_partitions, a->random_seed, shard_index, num_shards,
                                       w.d_sa, w.d_lcp, cap, &w.s, stats)
            : sufr_hip_sort_device_u64(ctx, w.d_text, n, flags, mql, a->seed_mask, a->num_partitions, a->random_seed, shard_index, num_shards,
                                       w.d_sa, w.d_lcp, cap, &w.s, stats);
        if (rc != SUFR_HIP_E_CAPACITY || w.s <= cap) break;
        (void)hipFree(w.d_sa); (void)hipFree(w.d_lcp); w.d_sa = w.d_lcp = nullptr;
        cap = w.s;                                   // (the call reports what it needed)
    }
    if (rc) { wide_free(w); return rc; }
    // the raw text is not read again: the stitch and the merge use the context's normalised copy, the file's text section comes
    // from the host (advisor r5: n bytes of HBM held until sufr_hip_shard_write for nothing)
    (void)hipFree(w.d_text); w.d_text = nullptr;
    info->num_suffixes = w.s;
    if (w.s) {
        uint64_t f = 0, l = 0;
        if (hipMemcpy(&f, w.d_sa, (size_t)w.width, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&l, (const uint8_t*)w.d_sa + (w.s - 1) * (size_t)w.width, (size_t)w.width, hipMemcpyDeviceToHost) != hipSuccess) {
            wide_free(w);
            sufr_hip_set_error_(ctx, "reading a shard's ends failed");
            return SUFR_HIP_E_HIP;
        }
        info->first_suffix = w.first = f; info->last_suffix = l;      // (little-endian: a 4-byte value lands in the low half)
    }
    std::lock_guard<std::mutex> g(g_wide_mu);
    g_wide[ctx] = w;
    return 0;
}

static int wide_shard_write(sufr_hip_ctx* ctx, const sufr_sequence_data* sd, const sufr_create_args* a, const char* outfile,
                            uint64_t num_suffixes, uint64_t total_suffixes, uint64_t suffix_offset, int has_prev, uint64_t prev_last_suffix,
                            int write_text)
{
    WideResident w;
    {
        std::lock_guard<std::mutex> g(g_wide_mu);
        auto it = g_wide.find(ctx);
        if (it == g_wide.end() || it->second.n != sd->seq_len || it->second.s != num_suffixes) {
            sufr_hip_set_error_(ctx, "sufr_hip_shard_write: the context holds no windowed shard of this text (sufr_hip_shard_build first; a write consumes it)");
            return SUFR_HIP_E_INVALID;
        }
        w = it->second;
        g_wide.erase(it);
    }
    struct Free { WideResident& w; ~Free() { wide_free(w); } } guard{w};
    const uint64_t n = sd->seq_len;
    int rc = 0;
    if (has_prev && num_suffixes) {
        // the boundary LCP under the order of the build (k_lcp_stitch): a two-row table {-, prev.last, 1}, {first, -, s}
        const uint64_t rows[6] = {0, prev_last_suffix, 1, w.first, 0, w.s};
        void* d_rows = nullptr;
        if (hipSetDevice(w.device) != hipSuccess || hipMalloc(&d_rows, sizeof rows) != hipSuccess ||
            hipMemcpy(d_rows, rows, sizeof rows, hipMemcpyHostToDevice) != hipSuccess) {
            if (d_rows) (void)hipFree(d_rows);
            sufr_hip_set_error_(ctx, "uploading the boundary of a windowed shard failed");
            return SUFR_HIP_E_HIP;
        }
        rc = w.width == 4 ? sufr_hip_stitch_device_u32(ctx, n, (const uint64_t*)d_rows, 1, 2, w.d_lcp)
                          : sufr_hip_stitch_device_u64(ctx, n, (const uint64_t*)d_rows, 1, 2, w.d_lcp);
        (void)hipFree(d_rows);
        if (rc) return rc;
    }
    const SufrLayout L = layout_of(*sd, a, total_suffixes);
    int fd = ::open(outfile, O_WRONLY);
    if (fd < 0) { sufr_hip_set_error_(ctx, (std::string(outfile) + ": " + strerror(errno)).c_str()); return SUFR_HIP_E_IO; }
    bool ok = true;
    if (write_text) {                                    // the normalised text, from the host
        const uint64_t PIECE = (uint64_t)32 << 20;
        std::vector<uint8_t> buf(PIECE);
        for (uint64_t o = 0; ok && o < n; o += PIECE) {
            const uint64_t len = n - o < PIECE ? n - o : PIECE;
            (void)sufr_hip_normalize(sd->seq + o, buf.data(), len, a->ignore_softmask);
            ok = pwrite_all(fd, buf.data(), len, L.text_pos + o);
        }
    }
    int failed = 0;
    if (ok && num_suffixes) {
        std::vector<Section> secs;
        secs.push_back({w.d_sa, num_suffixes * (uint64_t)w.width, L.sa_pos + suffix_offset * (uint64_t)w.width});
        secs.push_back({w.d_lcp, num_suffixes * (uint64_t)w.width, L.lcp_pos + suffix_offset * (uint64_t)w.width});
        failed = stream_sections(w.device, fd, secs);
    }
    if (close(fd) != 0 && !failed) failed = 2;
    if (!ok) failed = 2;
    if (failed) {
        sufr_hip_set_error_(ctx, failed == 2 ? (std::string(outfile) + ": write failed").c_str() : "device-to-host copy of the arrays failed");
        return failed == 2 ? SUFR_HIP_E_IO : SUFR_HIP_E_HIP;
    }
    return 0;
}

int sufr_hip_shard_build(sufr_hip_ctx* ctx, const sufr_sequence_data* sd, const sufr_create_args* a,
                         uint32_t shard_index, uint32_t num_shards, sufr_shard_info* info, sufr_hip_stats* stats)
{
    if (!ctx || !a || !sd || !sd->seq || !info) return SUFR_HIP_E_INVALID;
    memset(info, 0, sizeof *info);
    if (a->has_max_query_len && a->seed_mask) {                      // clap's conflicts_with; builder check 163-165
        sufr_hip_set_error_(ctx, "Cannot use max_query_len and seed_mask together");
        return SUFR_HIP_E_CONFLICT;
    }
    if (num_shards == 0 || shard_index >= num_shards) { sufr_hip_set_error_(ctx, "bad shard index"); return SUFR_HIP_E_INVALID; }
    if (sufr_hip_is_wide_(ctx, sd->seq_len))                       // a shard of the windowed build (ranges of the first 8 bytes)
        return wide_shard_build(ctx, sd, a, shard_index, num_shards, info, stats);
    sufr_io_release_ctx_(ctx);
    uint64_t s = 0;
    int rc = sufr_hip_build_resident_(ctx, sd->seq, sd->seq_len, build_flags(a),
                                      a->has_max_query_len ? a->max_query_len : 0, a->seed_mask, shard_index,
                                      num_shards, &s, stats, nullptr, nullptr, nullptr, nullptr);
    if (rc != 0) return rc;
    info->num_suffixes = s;
    if (s && (rc = sufr_hip_resident_ends_(ctx, s, &info->first_suffix, &info->last_suffix))) return rc;
    return 0;
}

int sufr_hip_shard_write(sufr_hip_ctx* ctx, const sufr_sequence_data* sd, const sufr_create_args* a,
                         const char* outfile, uint64_t num_suffixes, uint64_t total_suffixes, uint64_t suffix_offset,
                         int has_prev, uint64_t prev_last_suffix, int write_text)
{
    if (!ctx || !a || !sd || !outfile) return SUFR_HIP_E_INVALID;
    if (sufr_hip_is_wide_(ctx, sd->seq_len))
        return wide_shard_write(ctx, sd, a, outfile, num_suffixes, total_suffixes, suffix_offset, has_prev, prev_last_suffix, write_text);
    int rc;
    if (has_prev && num_suffixes && (rc = sufr_hip_resident_stitch_(ctx, sd->seq_len, prev_last_suffix, nullptr))) return rc;
    int device = 0;
    const void *d_text = nullptr, *d_sa = nullptr, *d_lcp = nullptr;
    if ((rc = sufr_hip_resident_arrays_(ctx, &device, &d_text, &d_sa, &d_lcp))) return rc;
    const SufrLayout L = layout_of(*sd, a, total_suffixes);
    int fd = ::open(outfile, O_WRONLY);
    if (fd < 0) { sufr_hip_set_error_(ctx, (std::string(outfile) + ": " + strerror(errno)).c_str()); return SUFR_HIP_E_IO; }
    std::vector<Section> secs;
    if (write_text) secs.push_back({d_text, sd->seq_len, L.text_pos});
    secs.push_back({d_sa, num_suffixes * 4, L.sa_pos + suffix_offset * 4});
    secs.push_back({d_lcp, num_suffixes * 4, L.lcp_pos + suffix_offset * 4});
    std::thread freer([ctx, write_text]() { sufr_hip_release_build_arrays_(ctx, !write_text); });     // (under the copies: not in the process's exit)
    int failed = stream_sections(device, fd, secs);
    freer.join();
    if (close(fd) != 0 && !failed) failed = 2;
    if (failed) {
        sufr_hip_set_error_(ctx, failed == 2 ? (std::string(outfile) + ": write failed").c_str()
                                             : "device-to-host copy of the arrays failed");
        return failed == 2 ? SUFR_HIP_E_IO : SUFR_HIP_E_HIP;
    }
    return 0;
}

// Texts that take windows (2^32 - 2^24 bytes and more, or above the caller's window) in K >= n_ctx shards (round 5): shard r of
// the windowed build (ranges of the first 8 bytes, sufr_wide.inc) is built on ctxs[r % n_ctx] into device arrays of the file's
// index width, its first LCP is stitched on its device, and its slice is streamed to its place in the one file; the contexts
// take the shards in rounds of n_ctx.  K > n_ctx is the OUT-OF-CORE form of the build: the suffix and LCP
// arrays of the whole text are never in HBM together -- a device holds the text, the windows' workspace and ONE shard's arrays,
// which leave for the file before the next shard is built (the reference's counterpart: partitions sorted one at a time out of
// temporary files, sufr_builder.rs:495-598, concatenated by write() 875-906).  The price is a windowed sort of the text per
// shard -- seconds -- against the minutes the file takes to write.  The section offsets need the number of suffixes before the
// first shard exists: it is counted on the host while the text section is normalised and written (eligibility 446-449 is a
// property of the normalised byte), and checked against the shards' sum at the end.
static int create_wide_sharded(sufr_hip_ctx* const* ctxs, int n_ctx, uint32_t K, const sufr_sequence_data& sd, const sufr_create_args* a,
                               const std::string& outfile, sufr_hip_stats* stats)
{
    sufr_hip_ctx* ctx0 = ctxs[0];
    const uint64_t n = sd.seq_len;
    const int width = n < 0xFFFFFFFFull ? 4 : 8;               // suffix_array.rs:461
    const uint32_t flags = build_flags(a);
    const uint64_t mql = a->has_max_query_len ? a->max_query_len : 0;
    struct Dev { void *d_text = nullptr, *d_sa = nullptr, *d_lcp = nullptr, *d_bounds = nullptr; uint64_t cap = 0, s = 0; int rc = 0, device = 0; sufr_hip_stats st; };
    std::vector<Dev> dv(n_ctx);
    auto release = [&]() {
        for (Dev& x : dv) {
            if (hipSetDevice(x.device) != hipSuccess) continue;
            for (void* q : {x.d_text, x.d_sa, x.d_lcp, x.d_bounds}) if (q) (void)hipFree(q);
            x.d_text = x.d_sa = x.d_lcp = x.d_bounds = nullptr;
        }
    };
    const double t0 = now_s();
    // ---- the file: header, name table, and the normalised text from the host (counting the suffixes on the way) ----
    struct stat ost;
    const int lrc = lstat(outfile.c_str(), &ost);
    const bool in_place = lrc == 0 ? !(S_ISREG(ost.st_mode) && ost.st_nlink == 1) : errno != ENOENT;      // (see sufr_hip_create_from_sequence_multi)
    const std::string partial = in_place ? outfile : outfile + ".partial";
    int fd = ::open(partial.c_str(), O_WRONLY | O_CREAT | O_TRUNC, lrc == 0 && !in_place ? (ost.st_mode & 07777) : 0644);
    if (fd < 0) { sufr_hip_set_error_(ctx0, (outfile + ": " + strerror(errno)).c_str()); return SUFR_HIP_E_IO; }
    auto fail = [&](int rc, const char* msg) -> int {
        if (fd >= 0) (void)close(fd);
        if (!in_place) (void)unlink(partial.c_str());
        release();
        if (msg) sufr_hip_set_error_(ctx0, msg);
        return rc;
    };
    std::atomic<uint64_t> eligible{0};
    {
        // (the text's place does not depend on the suffix count -- but on the index width: the sequence starts before it are T-wide)
        const uint64_t text_pos = sufr_layout(a->is_dna, a->allow_ambiguity, a->ignore_softmask, n, width, 0, a->has_max_query_len, a->max_query_len,
                                              a->seed_mask, sd.start_positions, sd.num_sequences, (const char* const*)sd.sequence_names).text_pos;
        const uint64_t PIECE = (uint64_t)32 << 20;
        const uint64_t npieces = (n + PIECE - 1) / PIECE;
        std::atomic<uint64_t> next{0};
        std::atomic<int> bad{0};
        unsigned W = host_threads(8);
        if (W > npieces) W = (unsigned)npieces;
        const bool every = !a->is_dna || a->allow_ambiguity;
        std::vector<std::thread> th;
        for (unsigned w = 0; w < W; w++)
            th.emplace_back([&]() {
                std::vector<uint8_t> buf(PIECE);
                uint64_t mine = 0;
                for (uint64_t i; !bad && (i = next.fetch_add(1)) < npieces;) {
                    const uint64_t o = i * PIECE, len = n - o < PIECE ? n - o : PIECE;
                    (void)sufr_hip_normalize(sd.seq + o, buf.data(), len, a->ignore_softmask);
                    if (every) mine += len;
                    else for (uint64_t k = 0; k < len; k++) { const uint8_t c = buf[k]; mine += (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T') | (c == '$'); }
                    if (!pwrite_all(fd, buf.data(), len, text_pos + o)) bad = 1;
                }
                eligible += mine;
            });
        for (auto& t : th) t.join();
        if (bad) return fail(SUFR_HIP_E_IO, (outfile + ": write failed").c_str());
    }
    const uint64_t total = eligible.load();
    const SufrLayout L = sufr_layout(a->is_dna, a->allow_ambiguity, a->ignore_softmask, n, width, total, a->has_max_query_len, a->max_query_len,
                                     a->seed_mask, sd.start_positions, sd.num_sequences, (const char* const*)sd.sequence_names);
    if (!pwrite_all(fd, L.head.data(), L.head.size(), 0) || !pwrite_all(fd, L.tail.data(), L.tail.size(), L.tail_pos))
        return fail(SUFR_HIP_E_IO, (outfile + ": write failed").c_str());
    // ---- the text, once per context ----
    {
        std::vector<std::thread> th;
        for (int c = 0; c < n_ctx; c++)
            th.emplace_back([&, c]() {
                Dev& x = dv[c];
                memset(&x.st, 0, sizeof x.st);
                x.device = sufr_hip_ctx_device_(ctxs[c]);
                if (hipSetDevice(x.device) != hipSuccess || hipMalloc(&x.d_text, n + 64) != hipSuccess ||
                    hipMemcpy(x.d_text, sd.seq, n, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMalloc(&x.d_bounds, (size_t)K * 24) != hipSuccess) {
                    sufr_hip_set_error_(ctxs[c], "out of device memory (text of a windowed shard)"); x.rc = SUFR_HIP_E_NOMEM;
                }
            });
        for (auto& t : th) t.join();
        for (int c = 0; c < n_ctx; c++)
            if (dv[c].rc) { if (c) sufr_hip_set_error_(ctx0, sufr_hip_last_error(ctxs[c])); return fail(dv[c].rc, nullptr); }
    }
    // ---- the shards, in rounds of n_ctx ----
    std::vector<uint64_t> bounds((size_t)K * 3, 0);            // {first suffix, last suffix, count} of every shard built so far
    uint64_t off = 0;
    double t_sort = 0.0;
    for (uint32_t r0 = 0; r0 < K; r0 += (uint32_t)n_ctx) {
        const int live = (int)(K - r0 < (uint32_t)n_ctx ? K - r0 : (uint32_t)n_ctx);
        const double tb = now_s();
        std::vector<std::thread> th;
        for (int c = 0; c < live; c++)
            th.emplace_back([&, c]() {
                Dev& x = dv[c];
                const uint32_t r = r0 + (uint32_t)c;
                if (hipSetDevice(x.device) != hipSuccess) { x.rc = SUFR_HIP_E_HIP; return; }
                // a shard holds ~1 / K of the suffixes (quantiles of a sample); one that turns out larger gets its size in a second try
                uint64_t want = total / K + total / (4 * (uint64_t)K) + ((uint64_t)1 << 20);
                if (want > total) want = total;
                if (want < 1) want = 1;
                for (int attempt = 0; attempt < 2; attempt++) {
                    if (x.cap < want) {
                        if (x.d_sa) (void)hipFree(x.d_sa);
                        if (x.d_lcp) (void)hipFree(x.d_lcp);
                        x.d_sa = x.d_lcp = nullptr; x.cap = 0;
                        if (hipMalloc(&x.d_sa, want * (size_t)width + 16) != hipSuccess || hipMalloc(&x.d_lcp, want * (size_t)width + 16) != hipSuccess) {
                            sufr_hip_set_error_(ctxs[c], "out of device memory (arrays of a windowed shard)"); x.rc = SUFR_HIP_E_NOMEM; return;
                        }
                        x.cap = want;
                    }
                    sufr_hip_stats one;
                    memset(&one, 0, sizeof one);
                    x.s = 0;
                    x.rc = width == 4
                        ? sufr_hip_sort_device_u32(ctxs[c], x.d_text, n, flags, mql, a->seed_mask, a->num_partitions, a->random_seed,
                                                   r, K, x.d_sa, x.d_lcp, x.cap, &x.s, &one)
                        : sufr_hip_sort_device_u64(ctxs[c], x.d_text, n, flags, mql, a->seed_mask, a->num_partitions, a->random_seed,
                                                   r, K, x.d_sa, x.d_lcp, x.cap, &x.s, &one);
                    if (r0 == 0) x.st = one;
                    else { x.st.ms_total += one.ms_total; x.st.ms_partition += one.ms_partition; x.st.ms_passes += one.ms_passes; x.st.ms_deep += one.ms_deep;
                           x.st.ms_normalize += one.ms_normalize; x.st.ms_hist_text += one.ms_hist_text; x.st.ms_finish += one.ms_finish;
                           x.st.deep_records += one.deep_records; x.st.num_suffixes += one.num_suffixes; }
                    if (x.rc != SUFR_HIP_E_CAPACITY || x.s <= x.cap) break;
                    want = x.s;                          // (the call reports what it needed)
                }
                if (x.rc == 0 && x.s) {
                    uint64_t f = 0, l = 0;
                    if (hipMemcpy(&f, x.d_sa, (size_t)width, hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(&l, (const uint8_t*)x.d_sa + (x.s - 1) * (size_t)width, (size_t)width, hipMemcpyDeviceToHost) != hipSuccess) {
                        sufr_hip_set_error_(ctxs[c], "reading a shard's ends failed"); x.rc = SUFR_HIP_E_HIP; return;
                    }
                    bounds[3 * (size_t)r] = f; bounds[3 * (size_t)r + 1] = l;     // (little-endian: a 4-byte value lands in the low half)
                }
                bounds[3 * (size_t)r + 2] = x.s;
            });
        for (auto& t : th) t.join();
        for (int c = 0; c < live; c++)
            if (dv[c].rc) { if (c) sufr_hip_set_error_(ctx0, sufr_hip_last_error(ctxs[c])); return fail(dv[c].rc, nullptr); }
        t_sort += now_s() - tb;
        // the boundary LCP of every shard of the round but the globally first, under the order of the build (k_lcp_stitch), then
        // the slices
        for (int c = 0; c < live; c++) {
            Dev& x = dv[c];
            const uint32_t r = r0 + (uint32_t)c;
            if (!x.s) continue;
            if (off + x.s > total) return fail(SUFR_HIP_E_HIP, "the shards hold more suffixes than the text has eligible positions (internal error)");
            if (r > 0) {
                hipError_t he = hipSetDevice(x.device);
                if (he == hipSuccess) he = hipMemcpy(x.d_bounds, bounds.data(), bounds.size() * 8, hipMemcpyHostToDevice);
                if (he != hipSuccess) return fail(SUFR_HIP_E_HIP, (std::string("uploading the shard bounds failed: ") + hipGetErrorString(he)).c_str());
                const int rc = width == 4 ? sufr_hip_stitch_device_u32(ctxs[c], n, (const uint64_t*)x.d_bounds, r, K, x.d_lcp)
                                          : sufr_hip_stitch_device_u64(ctxs[c], n, (const uint64_t*)x.d_bounds, r, K, x.d_lcp);
                if (rc) {
                    const std::string why = std::string("stitching shard ") + std::to_string(r) + " failed: " + sufr_hip_last_error(ctxs[c]);
                    return fail(rc, why.c_str());
                }
            }
            std::vector<Section> secs;
            secs.push_back({x.d_sa, x.s * (uint64_t)width, L.sa_pos + off * (uint64_t)width});
            secs.push_back({x.d_lcp, x.s * (uint64_t)width, L.lcp_pos + off * (uint64_t)width});
            const int failed = stream_sections(x.device, fd, secs);
            if (failed) return fail(failed == 1 ? SUFR_HIP_E_HIP : SUFR_HIP_E_IO, failed == 1 ? "device-to-host copy of the arrays failed" : (outfile + ": write failed").c_str());
            off += x.s;
        }
    }
    if (off != total)
        return fail(SUFR_HIP_E_HIP, ("the shards hold " + std::to_string(off) + " suffixes, the text has " + std::to_string(total) + " eligible positions (internal error)").c_str());
    const int crc = close(fd);
    fd = -1;
    if (crc != 0) return fail(SUFR_HIP_E_IO, (outfile + ": write failed").c_str());
    release();
    if (!in_place && rename(partial.c_str(), outfile.c_str()) != 0) {
        sufr_hip_set_error_(ctx0, (outfile + ": " + strerror(errno)).c_str());
        (void)unlink(partial.c_str());
        return SUFR_HIP_E_IO;
    }
    if (stats)
        for (int c = 0; c < n_ctx; c++) {
            stats[c] = dv[c].st;
            stats[c].host_read_s = 0.0f; stats[c].host_build_s = (float)t_sort; stats[c].host_write_s = (float)(now_s() - t0 - t_sort);
        }
    return 0;
}

// How many shards a windowed create takes: the contexts' count, or -- a budget for the arrays set on the first context
// (sufr_hip_set_array_budget) -- as many as keep one shard's SA + LCP within it.  Without a budget the count doubles while a
// build runs out of device memory (up to 64 shards per context): nothing has to be estimated about the windows' workspace.
static int create_wide_auto(sufr_hip_ctx* const* ctxs, int n_ctx, uint32_t K0, const sufr_sequence_data& sd, const sufr_create_args* a,
                            const std::string& outfile, sufr_hip_stats* stats)
{
    const uint64_t n = sd.seq_len;
    const uint64_t budget = sufr_hip_array_budget_(ctxs[0]);
    uint32_t K = K0 < (uint32_t)n_ctx ? (uint32_t)n_ctx : K0;
    if (budget) {
        const uint64_t bytes = 2 * n * (uint64_t)(n < 0xFFFFFFFFull ? 4 : 8);      // (an upper bound: every position a suffix)
        uint64_t k = (bytes + budget - 1) / budget;
        if (k > 4096) k = 4096;
        if (k > K) K = (uint32_t)k;
    }
    const bool say = getenv("SUFR_HIP_DEBUG") != nullptr;
    for (;;) {
        const double t0 = now_s();
        const int rc = create_wide_sharded(ctxs, n_ctx, K, sd, a, outfile, stats);
        if (say) fprintf(stderr, "[sufr_hip] windowed create: %u shard%s over %d context%s (array budget %llu): rc %d in %.2f s\n", K, K == 1 ? "" : "s", n_ctx,
                         n_ctx == 1 ? "" : "s", (unsigned long long)budget, rc, now_s() - t0);
        if (rc != SUFR_HIP_E_NOMEM || K >= 64u * (uint32_t)n_ctx) return rc;
        // the failed attempt's window arrays are sized for K shards and never shrink by themselves: handed back, so that the
        // retry has what a fresh build with 2 K shards would have (advisor r5)
        for (int c = 0; c < n_ctx; c++) sufr_hip_release_wide_arrays_(ctxs[c]);
        K *= 2;
    }
}

// One process, several GPUs: shard r of n_ctx is built on ctxs[r] (a thread per context), then every context
// streams its SA / LCP slice to  sa_pos + 4 * (suffixes of the shards before it)  -- the multi-writer form of
// SufrBuilder::write (sufr_builder.rs:875-906); the first LCP of every shard but the first is the boundary fix
// 893-902, computed on that shard's device.  Contexts may share a device (tests: N shards on one GPU).
int sufr_hip_create_from_sequence_multi(sufr_hip_ctx* const* ctxs, int n_ctx, const sufr_sequence_data* sdp,
                                        const sufr_create_args* a, char* path_out, size_t path_out_len,
                                        sufr_hip_stats* stats)
{
    if (!ctxs || n_ctx < 1 || !a || !a->input || !sdp || !sdp->seq) return SUFR_HIP_E_INVALID;
    for (int r = 0; r < n_ctx; r++) if (!ctxs[r]) return SUFR_HIP_E_INVALID;
    sufr_hip_ctx* ctx0 = ctxs[0];
    const sufr_sequence_data& sd = *sdp;
    const std::string outfile = output_name(a);
    if (path_out && path_out_len) snprintf(path_out, path_out_len, "%s", outfile.c_str());
    if (sufr_hip_is_wide_(ctx0, sd.seq_len)) {
        // windowed build: its shards over the contexts (round 5); a seed mask or a cap below 8 symbols is not sharded in windows,
        // and one context takes the host-buffer path
        bool all_wide = true;
        for (int r = 0; r < n_ctx; r++) all_wide = all_wide && sufr_hip_is_wide_(ctxs[r], sd.seq_len);
        if (n_ctx == 1 || !all_wide || a->seed_mask || (a->has_max_query_len && a->max_query_len < 8))
            return sufr_hip_create_from_sequence(ctx0, sdp, a, path_out, path_out_len, stats);
        if (a->has_max_query_len && a->seed_mask) { sufr_hip_set_error_(ctx0, "Cannot use max_query_len and seed_mask together"); return SUFR_HIP_E_CONFLICT; }
        return create_wide_auto(ctxs, n_ctx, (uint32_t)n_ctx, sd, a, outfile, stats);
    }
    if (a->has_max_query_len && a->max_query_len < 8) n_ctx = 1;   // (a cap shorter than a first digit ties suffixes across shards)
    if (a->has_max_query_len && a->seed_mask) {            // clap's conflicts_with; builder check 163-165 (before the file is touched)
        sufr_hip_set_error_(ctx0, "Cannot use max_query_len and seed_mask together");
        return SUFR_HIP_E_CONFLICT;
    }
    const double t0 = now_s();
    // The text section (the normalised text, sufr_builder.rs:871) does not wait for the build: its place in the file is
    // known from the header alone, and the text map 144-160 is a byte table -- host threads map 32 MB pieces and write them
    // while the GPUs partition and sort (3.1 of the 15.1 GB of a human-sized file, off the D2H + write phase).
    // Written under "<output>.partial" and renamed at the end -- a failed build leaves an existing file alone -- but ONLY when
    // the output does not exist yet or is a plain file with one name: a rename would replace a symlink instead of writing
    // through it, drop the other names of a hard-linked file, and put a 15 GB regular file where /dev/null or a FIFO was
    // (`sufr create -o /dev/null` is a common way to time a build).  Anything else is opened in place, as the reference's
    // File::create does (sufr_builder.rs:819), and never unlinked.
    struct stat ost;
    const int lrc = lstat(outfile.c_str(), &ost);
    const bool in_place = lrc == 0 ? !(S_ISREG(ost.st_mode) && ost.st_nlink == 1) : errno != ENOENT;
    const std::string partial = in_place ? outfile : outfile + ".partial";
    auto discard = [&]() { if (!in_place) (void)unlink(partial.c_str()); };
    int tfd = ::open(partial.c_str(), O_WRONLY | O_CREAT | O_TRUNC, lrc == 0 && !in_place ? (ost.st_mode & 07777) : 0644);
    if (tfd < 0) { sufr_hip_set_error_(ctx0, (outfile + ": " + strerror(errno)).c_str()); return SUFR_HIP_E_IO; }
    if (lrc == 0 && !in_place) (void)fchmod(tfd, ost.st_mode & 07777);     // (the umask does not get a say over an existing file's mode)
    std::atomic<int> text_failed{0};
    std::vector<std::thread> text_writers;
    {
        const uint64_t text_pos = layout_of(sd, a, 0).text_pos;
        const uint64_t PIECE = (uint64_t)32 << 20;
        const uint64_t npieces = (sd.seq_len + PIECE - 1) / PIECE;
        auto next = std::make_shared<std::atomic<uint64_t>>(0);
        unsigned W = host_threads(8);
        if (W > npieces) W = (unsigned)npieces;
        const int soft = a->ignore_softmask;
        for (unsigned w = 0; w < W; w++)
            text_writers.emplace_back([&sd, &text_failed, tfd, text_pos, PIECE, npieces, next, soft]() {
                std::vector<uint8_t> buf(PIECE);
                for (uint64_t i; !text_failed && (i = next->fetch_add(1)) < npieces;) {
                    const uint64_t o = i * PIECE, len = sd.seq_len - o < PIECE ? sd.seq_len - o : PIECE;
                    (void)sufr_hip_normalize(sd.seq + o, buf.data(), len, soft);
                    if (!pwrite_all(tfd, buf.data(), len, text_pos + o)) text_failed = 1;
                }
            });
    }
    auto finish_text = [&]() -> bool {
        for (auto& x : text_writers) x.join();
        text_writers.clear();
        const bool ok = close(tfd) == 0 && !text_failed;
        tfd = -1;
        return ok;
    };
    std::vector<sufr_shard_info> info(n_ctx);
    std::vector<sufr_hip_stats> st(n_ctx);
    std::vector<int> rcs(n_ctx, 0);
    {
        std::vector<std::thread> th;
        for (int r = 0; r < n_ctx; r++)
            th.emplace_back([&, r]() { rcs[r] = sufr_hip_shard_build(ctxs[r], &sd, a, (uint32_t)r, (uint32_t)n_ctx, &info[r], &st[r]); });
        for (auto& x : th) x.join();
    }
    for (int r = 0; r < n_ctx; r++)
        if (rcs[r]) {
            (void)finish_text(); discard();
            if (r) sufr_hip_set_error_(ctx0, sufr_hip_last_error(ctxs[r]));
            return rcs[r];
        }
    const double t_built = now_s();
    uint64_t total = 0;
    std::vector<uint64_t> off(n_ctx, 0);
    for (int r = 0; r < n_ctx; r++) { off[r] = total; total += info[r].num_suffixes; }
    int rc = 0;
    {
        const SufrLayout L = layout_of(sd, a, total);      // header and name table around the sections (no O_TRUNC: the text is landing)
        if (!pwrite_all(tfd, L.head.data(), L.head.size(), 0) || !pwrite_all(tfd, L.tail.data(), L.tail.size(), L.tail_pos)) rc = SUFR_HIP_E_IO;
    }
    if (rc != 0) {
        (void)finish_text(); discard();
        sufr_hip_set_error_(ctx0, (outfile + ": write failed").c_str());
        return rc;
    }
    {
        std::vector<std::thread> th;
        int prev = -1;                                   // last non-empty shard before r
        for (int r = 0; r < n_ctx; r++) {
            const int pv = prev;
            th.emplace_back([&, r, pv]() {
                rcs[r] = sufr_hip_shard_write(ctxs[r], &sd, a, partial.c_str(), info[r].num_suffixes, total, off[r],
                                              pv >= 0, pv >= 0 ? info[pv].last_suffix : 0, 0);
            });
            if (info[r].num_suffixes) prev = r;
        }
        for (auto& x : th) x.join();
    }
    if (!finish_text()) {
        discard();
        sufr_hip_set_error_(ctx0, (outfile + ": write failed").c_str());
        return SUFR_HIP_E_IO;
    }
    for (int r = 0; r < n_ctx; r++)
        if (rcs[r]) {
            discard();                                   // no valid header over missing sections left behind
            if (r) sufr_hip_set_error_(ctx0, sufr_hip_last_error(ctxs[r]));
            return rcs[r];
        }
    if (!in_place && rename(partial.c_str(), outfile.c_str()) != 0) {
        sufr_hip_set_error_(ctx0, (outfile + ": " + strerror(errno)).c_str());
        discard();
        return SUFR_HIP_E_IO;
    }
    if (stats) {
        for (int r = 0; r < n_ctx; r++) {
            stats[r] = st[r];
            stats[r].host_read_s = 0.0f;                 // filled in by the callers that read the file
            stats[r].host_build_s = (float)(t_built - t0);
            stats[r].host_write_s = (float)(now_s() - t_built);
        }
    }
    return 0;
}

int sufr_hip_create_from_sequence(sufr_hip_ctx* ctx, const sufr_sequence_data* sdp, const sufr_create_args* a,
                                  char* path_out, size_t path_out_len, sufr_hip_stats* stats)
{
    if (!ctx || !a || !a->input || !sdp || !sdp->seq) return SUFR_HIP_E_INVALID;
    const sufr_sequence_data& sd = *sdp;
    if (!sufr_hip_is_wide_(ctx, sd.seq_len)) {
        // one 32-bit window: SA, LCP and the normalised text stay in HBM after the build and are streamed to the file
        return sufr_hip_create_from_sequence_multi(&ctx, 1, sdp, a, path_out, path_out_len, stats);
    }
    // windowed build (sufr_wide.inc), u64 arrays from 2^32 - 1 bytes on (suffix_array.rs:461): host buffers
    char err[512] = {0};
    const std::string outfile = output_name(a);
    if (path_out && path_out_len) snprintf(path_out, path_out_len, "%s", outfile.c_str());
    if (a->has_max_query_len && a->seed_mask) {                      // clap's conflicts_with; builder check 163-165
        sufr_hip_set_error_(ctx, "Cannot use max_query_len and seed_mask together");
        return SUFR_HIP_E_CONFLICT;
    }
    const uint64_t n = sd.seq_len;
    const int width = n < 0xFFFFFFFFull ? 4 : 8;
    // Out of core: with a budget for the arrays (sufr_hip_set_array_budget), or when the whole arrays do not fit
    // the device beside the windows' workspace, the build runs shard after shard and every shard's slice leaves for the file
    // before the next one is built (create_wide_sharded).  Seed masks and caps below 8 symbols are not sharded: they keep the
    // whole arrays, through host buffers (below).
    // One shard (the whole arrays, when they fit) goes the same way: device arrays of the file's width streamed through pinned
    // buffers -- 14.5 s for the 4.4e9-byte text against 25.4 s through pageable host arrays and fwrite (round 5).
    const bool can_shard = !a->seed_mask && !(a->has_max_query_len && a->max_query_len < 8);
    if (can_shard) return create_wide_auto(&ctx, 1, 1, sd, a, outfile, stats);
    std::vector<uint8_t> norm;
    void *sa = nullptr, *lcp = nullptr;
    try { norm.resize(n); } catch (const std::bad_alloc&) { norm.clear(); }
    if (norm.size() == n) { sa = malloc((size_t)n * (size_t)width + 8); lcp = malloc((size_t)n * (size_t)width + 8); }
    if (!sa || !lcp) {
        free(sa); free(lcp);
        sufr_hip_set_error_(ctx, "out of host memory (arrays of a windowed build)");
        return SUFR_HIP_E_NOMEM;
    }
    uint64_t s = 0;
    const double t_build = now_s();
    int rc = width == 4
        ? sufr_hip_build_u32(ctx, sd.seq, n, build_flags(a), a->has_max_query_len ? a->max_query_len : 0, a->seed_mask,
                             a->num_partitions, a->random_seed, norm.data(), (uint32_t*)sa, (uint32_t*)lcp, n, &s, stats)
        : sufr_hip_build_u64(ctx, sd.seq, n, build_flags(a), a->has_max_query_len ? a->max_query_len : 0, a->seed_mask,
                             a->num_partitions, a->random_seed, norm.data(), (uint64_t*)sa, (uint64_t*)lcp, n, &s, stats);
    if (stats) stats->host_build_s = (float)(now_s() - t_build);
    if (rc == 0) {
        rc = sufr_write_file(outfile.c_str(), a->is_dna, a->allow_ambiguity, a->ignore_softmask, norm.data(), n,
                             width, sa, lcp, s, a->has_max_query_len, a->max_query_len, a->seed_mask,
                             sd.start_positions, sd.num_sequences, (const char* const*)sd.sequence_names, err,
                             sizeof err);
        if (rc != 0) sufr_hip_set_error_(ctx, err);
    }
    free(sa); free(lcp);
    return rc;
}


int sufr_hip_create_file(sufr_hip_ctx* ctx, const sufr_create_args* a, char* path_out, size_t path_out_len,
                         sufr_hip_stats* stats)
{
    if (!ctx || !a || !a->input) return SUFR_HIP_E_INVALID;
    char err[512] = {0};
    sufr_sequence_data sd;
    const double t_start = now_s();
    int rc = sufr_read_sequence_file(a->input, a->sequence_delimiter ? a->sequence_delimiter : (uint8_t)'%', &sd,
                                     err, sizeof err);
    if (rc != 0) { sufr_hip_set_error_(ctx, err); return rc; }
    const double t_read = now_s();
    rc = sufr_hip_create_from_sequence(ctx, &sd, a, path_out, path_out_len, stats);
    if (stats) stats->host_read_s = (float)(t_read - t_start);
    sufr_sequence_data_free(&sd);
    return rc;
}

int sufr_hip_create_file_multi(sufr_hip_ctx* const* ctxs, int n_ctx, const sufr_create_args* a, char* path_out,
                               size_t path_out_len, sufr_hip_stats* stats)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0] || !a || !a->input) return SUFR_HIP_E_INVALID;
    char err[512] = {0};
    sufr_sequence_data sd;
    const double t_start = now_s();
    int rc = sufr_read_sequence_file(a->input, a->sequence_delimiter ? a->sequence_delimiter : (uint8_t)'%', &sd,
                                     err, sizeof err);
    if (rc != 0) { sufr_hip_set_error_(ctxs[0], err); return rc; }
    const double t_read = now_s();
    rc = sufr_hip_create_from_sequence_multi(ctxs, n_ctx, &sd, a, path_out, path_out_len, stats);
    if (stats) stats[0].host_read_s = (float)(t_read - t_start);
    sufr_sequence_data_free(&sd);
    return rc;
}

}  // extern "C"
