// sufr_cli.cpp -- `sufr create` on the MI355X build path.
//
// Command-line contract of the reference for this path: sufr/src/lib.rs:29-46 (global -t/--threads,
// -l/--log, --log-file), 83-125 (CreateArgs, alias `cr`), sufr/src/main.rs:8-40 (errors are printed as
// "Error: <msg>" and exit code 1).  The query sub-commands (count / locate / extract / list / summarize,
// sufr/src/lib.rs:129-271 for their arguments, 292-646 for their output) read the file through
// include/sufr_query.h; their output is the reference's, character for character.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <chrono>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/sufr_hip.h"
#include "../../include/sufr_query.h"

#include <algorithm>
#include <fstream>
#include <sstream>
#include <time.h>

namespace {

struct Log {
    int level = 0;  // 0 off, 1 info, 2 debug
    FILE* out = stdout;
    void info(const std::string& s) const { if (level >= 1) { fprintf(out, "[INFO  sufr] %s\n", s.c_str()); fflush(out); } }
    void debug(const std::string& s) const { if (level >= 2) { fprintf(out, "[DEBUG sufr] %s\n", s.c_str()); fflush(out); } }
};

std::string with_commas(uint64_t v)
{
    std::string s = std::to_string(v), r;
    int c = 0;
    for (size_t i = s.size(); i-- > 0;) { r.insert(r.begin(), s[i]); if (++c % 3 == 0 && i) r.insert(r.begin(), ','); }
    return r;
}

int usage(FILE* f)
{
    fprintf(f,
            "Usage: sufr [OPTIONS] <COMMAND> ...\n\n"
            "Commands:\n"
            "  create|cr    <INPUT>                 Create sufr file (on the GPU)\n"
            "  count|co     <SUFR> <QUERY>...       Count occurrences of sequences [-m LEN] [-o OUT] [-l] [-v]\n"
            "  locate|lo    <SUFR> <QUERY>...       Locate sequences [-a] [-m LEN] [-o OUT] [-l] [-v]\n"
            "  extract|ex   <SUFR> <QUERY>...       Extract sequences [-p PREFIX_LEN] [-s SUFFIX_LEN] [-m LEN] [-o OUT] [-l] [-v]\n"
            "  list|ls      <FILE> [RANK]...        List the suffix array [-r] [-s] [-p] [--len LEN] [-n NUM] [-o OUT] [-v]\n"
            "  summarize|su <SUFR>                  Summarize sufr file\n"
            "  count / locate / extract take --device <ID>: the queries are searched as one batch on that GPU\n\n"
            "Global options:\n"
            "  -t, --threads <THREADS>   Host workers of count / locate / extract [default: one per core]; create runs on the GPU\n"
            "  -l, --log <LOG>           Log level [possible values: info, debug]\n"
            "      --log-file <FILE>     Log file\n"
            "      --device <ID>         HIP device ordinal [default: 0]\n"
            "      --devices <ID,ID,...> Build on several GPUs: the suffixes are split by first-digit range, every GPU\n"
            "                            sorts its range and writes its slice of the one output file\n\n"
            "create options:\n"
            "  -n, --num-partitions <NUM_PARTS>  Subproblem count [default: 16]\n"
            "  -m, --max-query-len <CONTEXT>     Max context\n"
            "  -o, --output <OUTPUT>             Output file\n"
            "  -d, --dna                         Input is DNA\n"
            "  -a, --allow-ambiguity             Allow suffixes starting with ambiguity codes\n"
            "  -i, --ignore-softmask             Ignore suffixes in soft-mask/lowercase regions\n"
            "  -D, --sequence-delimiter <DELIM>  Character to separate sequences [default: %%]\n"
            "  -s, --seed-mask <MASK>            Spaced seeds mask\n"
            "  -r, --random-seed <RANDSEED>      Random seed [default: 42]\n"
            "      --window <POSITIONS>          Build texts longer than this in overlapping windows merged on the device\n"
            "                                    [default: one window below 2^32 - 2^24 bytes, as few as fit above]\n"
            "      --margin <POSITIONS>          Comparison context after every window [default: 2^26]\n"
            "      --array-budget <BYTES>        Windowed builds: device memory the suffix + LCP arrays may take at once; the build\n"
            "                                    runs shard after shard and streams every slice to the file [default: no limit;\n"
            "                                    arrays that do not fit are split by themselves]\n\n"
            "Texts of 2^32 - 2^24 bytes and more are built in overlapping 32-bit windows merged on the device; with --devices every\n"
            "device builds its shard of the windows (a seed mask: on the first device).\n");
    return f == stderr ? 2 : 0;
}


// ---------------------------------------------------------------------------------------------------------------
// query sub-commands (sufr/src/lib.rs:292-646)
// ---------------------------------------------------------------------------------------------------------------
struct QueryArgs {
    std::string file, output;
    std::vector<std::string> positional;        // queries / ranks
    bool has_mql = false; uint64_t mql = 0;
    bool abs = false;
    bool has_prefix = false, has_suffix = false; uint64_t prefix_len = 0, suffix_len = 0;
    bool show_rank = false, show_suffix = false, show_lcp = false;
    bool has_len = false, has_number = false; uint64_t len = 0, number = 0;
    int device = -1;                            // --device N: the whole batch of queries is searched on that GPU
    int threads = 0;                            // -t/--threads (global option, sufr/src/lib.rs:29-46): host search workers
};

// parse_locate_queries (lib.rs:449-466): an argument that names an existing file is read as whitespace-separated queries
std::vector<std::string> expand_queries(const std::vector<std::string>& args)
{
    std::vector<std::string> out;
    for (const std::string& a : args) {
        struct stat sb;
        if (stat(a.c_str(), &sb) == 0) {
            // an argument that names something on disk is a file of queries (sufr/src/lib.rs: the path is opened and
            // read; a directory or an unreadable file is an error there too, not an empty query list)
            std::ifstream in(a);
            if (!S_ISREG(sb.st_mode) || !in) {
                fprintf(stderr, "Error: %s: %s\n", a.c_str(), S_ISDIR(sb.st_mode) ? "Is a directory" : "cannot read the query file");
                exit(1);
            }
            std::string w;
            while (in >> w) out.push_back(w);
        } else out.push_back(a);
    }
    return out;
}

struct OutFile {
    FILE* f = stdout;
    bool open(const std::string& name)
    {
        if (name.empty()) return true;
        f = fopen(name.c_str(), "w");
        return f != nullptr;
    }
    ~OutFile() { if (f && f != stdout) fclose(f); }
};

sufr_file* open_or_die(const std::string& path)
{
    char err[512] = {0};
    sufr_file* f = nullptr;
    if (sufr_file_open(path.c_str(), &f, err, sizeof err) != 0) { fprintf(stderr, "Error: %s\n", err); exit(1); }
    return f;
}

// Rank range of every query: one by one on the host, or as one batch on a GPU (text + suffix array copied to HBM first:
// worth it for many queries; the answers are the same, tests/test_gpu_query.py)
struct Hit { bool found; uint64_t lo, hi; };
std::vector<Hit> search_all(sufr_file* f, const QueryArgs& a, const std::vector<std::string>& queries)
{
    std::vector<Hit> hits(queries.size(), Hit{false, 0, 0});
    std::string bytes;
    std::vector<uint64_t> off(queries.size() + 1, 0), lo(queries.size(), 0), hi(queries.size(), 0);
    for (size_t i = 0; i < queries.size(); i++) { bytes += queries[i]; off[i + 1] = bytes.size(); }
    if (a.device < 0) {
        sufr_file_search_batch(f, (const uint8_t*)bytes.data(), off.data(), queries.size(), a.has_mql, a.mql, lo.data(), hi.data(), a.threads);
        for (size_t i = 0; i < queries.size(); i++) hits[i] = Hit{hi[i] > lo[i], lo[i], hi[i]};
        return hits;
    }
    sufr_hip_ctx* ctx = sufr_hip_create(a.device);
    if (!ctx) { fprintf(stderr, "Error: %s\n", sufr_hip_last_error(nullptr)); exit(1); }
    sufr_hip_index* ix = nullptr;
    if (sufr_hip_index_load(ctx, f, &ix) != 0 ||
        sufr_hip_search_batch(ctx, ix, (const uint8_t*)bytes.data(), off.data(), queries.size(), a.has_mql, a.mql, lo.data(), hi.data()) != 0) {
        fprintf(stderr, "Error: %s\n", sufr_hip_last_error(ctx));
        exit(1);
    }
    for (size_t i = 0; i < queries.size(); i++) hits[i] = Hit{hi[i] > lo[i], lo[i], hi[i]};
    sufr_hip_index_free(ix);
    sufr_hip_destroy(ctx);
    return hits;
}

int cmd_count(const QueryArgs& a)
{
    sufr_file* f = open_or_die(a.file);
    OutFile out;
    if (!out.open(a.output)) { fprintf(stderr, "Error: %s: cannot create\n", a.output.c_str()); return 1; }
    const std::vector<std::string> queries = expand_queries(a.positional);
    const std::vector<Hit> hits = search_all(f, a, queries);
    for (size_t i = 0; i < queries.size(); i++)
        fprintf(out.f, "%s %llu\n", queries[i].c_str(), (unsigned long long)(hits[i].found ? hits[i].hi - hits[i].lo : 0));
    sufr_file_close(f);
    return 0;
}

int cmd_locate(const QueryArgs& a)
{
    sufr_file* f = open_or_die(a.file);
    OutFile out;
    if (!out.open(a.output)) { fprintf(stderr, "Error: %s: cannot create\n", a.output.c_str()); return 1; }
    const std::vector<std::string> queries = expand_queries(a.positional);
    const std::vector<Hit> hits = search_all(f, a, queries);
    for (size_t qi = 0; qi < queries.size(); qi++) {
        const std::string& q = queries[qi];
        const uint64_t lo = hits[qi].lo, hi = hits[qi].hi;
        if (!hits[qi].found) {
            fprintf(stderr, "%s not found\n", q.c_str());
            continue;
        }
        if (a.abs) {
            std::string line = q;
            for (uint64_t r = lo; r < hi; r++) line += " " + std::to_string(sufr_file_suffix(f, r));
            fprintf(out.f, "%s\n", line.c_str());
            continue;
        }
        // by sequence name, then position inside the sequence (lib.rs:513-518)
        struct Pos { std::string name; uint64_t at; };
        std::vector<Pos> ps;
        for (uint64_t r = lo; r < hi; r++) {
            const uint64_t sfx = sufr_file_suffix(f, r);
            const uint64_t i = sufr_file_sequence_of(f, sfx);
            ps.push_back({sufr_file_sequence_name(f, i), sfx - sufr_file_sequence_start(f, i)});
        }
        std::stable_sort(ps.begin(), ps.end(), [](const Pos& x, const Pos& y) { return x.name != y.name ? x.name < y.name : x.at < y.at; });
        fprintf(out.f, "%s\n", q.c_str());
        std::string prev, buf;
        for (const Pos& p : ps) {
            if (p.name != prev) {
                if (!buf.empty()) fprintf(out.f, "%s %s\n", prev.c_str(), buf.c_str());
                prev = p.name; buf.clear();
            }
            if (!buf.empty()) buf += ",";
            buf += std::to_string(p.at);
        }
        if (!buf.empty()) fprintf(out.f, "%s %s\n", prev.c_str(), buf.c_str());
        fprintf(out.f, "//\n");
    }
    sufr_file_close(f);
    return 0;
}

int cmd_extract(const QueryArgs& a)
{
    sufr_file* f = open_or_die(a.file);
    OutFile out;
    if (!out.open(a.output)) { fprintf(stderr, "Error: %s: cannot create\n", a.output.c_str()); return 1; }
    sufr_file_meta m; sufr_file_metadata(f, &m);
    const uint8_t* text = sufr_file_text(f);
    const std::vector<std::string> queries = expand_queries(a.positional);
    const std::vector<Hit> hits = search_all(f, a, queries);
    for (size_t qi = 0; qi < queries.size(); qi++) {
        const std::string& q = queries[qi];
        const uint64_t lo = hits[qi].lo, hi = hits[qi].hi;
        if (!hits[qi].found) {
            fprintf(stderr, "%s not found\n", q.c_str());
            continue;
        }
        for (uint64_t r = lo; r < hi; r++) {          // SufrFile::extract (sufr_file.rs:898-960)
            const uint64_t sfx = sufr_file_suffix(f, r);
            const uint64_t i = sufr_file_sequence_of(f, sfx);
            const uint64_t seq_start = sufr_file_sequence_start(f, i);
            const uint64_t seq_end = i + 1 == m.num_sequences ? m.text_len : sufr_file_sequence_start(f, i + 1);
            const uint64_t rel = sfx - seq_start;
            const uint64_t cstart = rel > (a.has_prefix ? a.prefix_len : 0) ? rel - (a.has_prefix ? a.prefix_len : 0) : 0;
            uint64_t cend = a.has_suffix ? rel + a.suffix_len : seq_end;
            if (cend > seq_end) cend = seq_end;
            // string_at(sequence_start + range.start, Some(range.end - range.start)) (sufr_file.rs:399-411)
            const uint64_t from = seq_start + cstart;
            uint64_t to = from + (cend > cstart ? cend - cstart : 0);
            if (to > m.text_len) to = m.text_len;
            fprintf(out.f, ">%s:%llu-%llu %s %llu\n", sufr_file_sequence_name(f, i), (unsigned long long)cstart,
                    (unsigned long long)cend, q.c_str(), (unsigned long long)(rel - cstart));
            fwrite(text + from, 1, (size_t)(to > from ? to - from : 0), out.f);
            fputc('\n', out.f);
        }
    }
    sufr_file_close(f);
    return 0;
}

// parse_index / parse_pos (lib.rs:556-590): "3", "1,5", "2-4" (inclusive)
bool parse_ranks(const std::string& arg, std::vector<uint64_t>& out, std::string& err)
{
    size_t p = 0;
    while (p <= arg.size()) {
        size_t q = arg.find(',', p);
        if (q == std::string::npos) q = arg.size();
        const std::string val = arg.substr(p, q - p);
        auto is_num = [](const std::string& s) { return !s.empty() && s.find_first_not_of("0123456789") == std::string::npos; };
        const size_t dash = val.find('-');
        if (is_num(val)) out.push_back(strtoull(val.c_str(), nullptr, 10));
        else if (dash != std::string::npos && is_num(val.substr(0, dash)) && is_num(val.substr(dash + 1))) {
            const uint64_t n1 = strtoull(val.substr(0, dash).c_str(), nullptr, 10), n2 = strtoull(val.substr(dash + 1).c_str(), nullptr, 10);
            if (n1 >= n2) {
                err = "First number in range (" + std::to_string(n1) + ") must be lower than second number (" + std::to_string(n2) + ")";
                return false;
            }
            for (uint64_t v = n1; v <= n2; v++) out.push_back(v);
        } else { err = "illegal list value: \"" + val + "\""; return false; }
        p = q + 1;
    }
    return true;
}

int cmd_list(const QueryArgs& a)
{
    sufr_file* f = open_or_die(a.file);
    OutFile out;
    if (!out.open(a.output)) { fprintf(stderr, "Error: %s: cannot create\n", a.output.c_str()); return 1; }
    std::vector<uint64_t> ranks;
    for (const std::string& r : a.positional) {
        std::string err;
        if (!parse_ranks(r, ranks, err)) { fprintf(stderr, "Error: %s\n", err.c_str()); sufr_file_close(f); return 1; }
    }
    sufr_file_meta m; sufr_file_metadata(f, &m);
    const uint8_t* text = sufr_file_text(f);
    const int width = (int)std::to_string(m.text_len).size();
    const uint64_t suffix_len = a.has_len ? a.len : m.text_len;
    auto print = [&](uint64_t rank) {                      // SufrFile::list (sufr_file.rs:1013-1077)
        const uint64_t sfx = sufr_file_suffix(f, rank);
        const uint64_t end = sfx + suffix_len > m.text_len ? m.text_len : sfx + suffix_len;
        if (a.show_rank) fprintf(out.f, "%*llu ", width, (unsigned long long)rank);
        if (a.show_suffix) fprintf(out.f, "%*llu ", width, (unsigned long long)sfx);
        if (a.show_lcp) fprintf(out.f, "%*llu ", width, (unsigned long long)sufr_file_lcp(f, rank));
        fwrite(text + sfx, 1, (size_t)(end - sfx), out.f);
        fputc('\n', out.f);
    };
    if (ranks.empty()) {
        const uint64_t number = a.has_number ? a.number : 0;
        for (uint64_t r = 0; r < m.len_suffixes; r++) {
            print(r);
            if (number > 0 && r == number - 1) break;
        }
    } else {
        for (uint64_t r : ranks) {
            if (r < m.len_suffixes) print(r);
            else fprintf(stderr, "Invalid rank: %llu\n", (unsigned long long)r);
        }
    }
    sufr_file_close(f);
    return 0;
}

// textwrap::wrap(s, width): greedy, breaks at spaces
std::vector<std::string> wrap_text(const std::string& s, size_t width)
{
    std::vector<std::string> lines;
    std::istringstream in(s);
    std::string w, cur;
    while (in >> w) {
        if (!cur.empty() && cur.size() + 1 + w.size() > width) { lines.push_back(cur); cur.clear(); }
        cur += (cur.empty() ? "" : " ") + w;
    }
    lines.push_back(cur);
    return lines;
}

int cmd_summarize(const QueryArgs& a)
{
    sufr_file* f = open_or_die(a.file);
    sufr_file_meta m; sufr_file_metadata(f, &m);
    std::vector<std::pair<std::string, std::vector<std::string>>> rows;     // name, value lines
    auto row = [&](const std::string& k, const std::string& v) { rows.push_back({k, {v}}); };
    row("Filename", a.file);
    {
        char buf[64];
        time_t t = (time_t)m.modified;
        struct tm tmv;
        localtime_r(&t, &tmv);
        strftime(buf, sizeof buf, "%Y-%m-%d %H:%M", &tmv);
        row("Modified", buf);
    }
    row("File Size", with_commas(m.file_size) + " bytes");
    row("File Version", std::to_string(m.version));
    row("DNA", m.is_dna ? "true" : "false");
    row("Allow Ambiguity", m.allow_ambiguity ? "true" : "false");
    row("Ignore Softmask", m.ignore_softmask ? "true" : "false");
    row("Text Length", with_commas(m.text_len));
    row("Len Suffixes", with_commas(m.len_suffixes));
    if (m.seed_mask_len) {
        std::string mask;
        const uint8_t* mb = sufr_file_seed_mask(f);
        for (uint64_t i = 0; i < m.seed_mask_len; i++) mask += mb[i] ? '1' : '0';
        row("Seed mask", mask);
    } else row("Max query len", with_commas(m.max_query_len));
    row("Num sequences", with_commas(m.num_sequences));
    std::string starts, names;
    for (uint64_t i = 0; i < m.num_sequences; i++) {
        starts += (i ? ", " : "") + std::to_string(sufr_file_sequence_start(f, i));
        names += std::string(i ? ", " : "") + sufr_file_sequence_name(f, i);
    }
    rows.push_back({"Sequence starts", wrap_text(starts, 40)});
    rows.push_back({"Sequence names", wrap_text(names, 40)});
    // tabled's default style: +---+---+ between rows, cells padded by one blank
    size_t w0 = 0, w1 = 0;
    for (auto& r : rows) { w0 = std::max(w0, r.first.size()); for (auto& l : r.second) w1 = std::max(w1, l.size()); }
    const std::string rule = "+" + std::string(w0 + 2, '-') + "+" + std::string(w1 + 2, '-') + "+";
    printf("%s\n", rule.c_str());
    for (auto& r : rows) {
        for (size_t i = 0; i < r.second.size(); i++)
            printf("| %-*s | %-*s |\n", (int)w0, i == 0 ? r.first.c_str() : "", (int)w1, r.second[i].c_str());
        printf("%s\n", rule.c_str());
    }
    sufr_file_close(f);
    return 0;
}

// arguments of the query sub-commands (clap definitions of lib.rs:129-271)
int run_query(const std::string& cmd, int argc, char** argv, int first, int threads)
{
    QueryArgs a;
    a.threads = threads;
    std::vector<std::string> pos;
    auto need = [&](int& i, const char* opt) -> const char* {
        if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", opt); exit(2); }
        return argv[++i];
    };
    const bool is_list = cmd == "list", is_extract = cmd == "extract", is_locate = cmd == "locate", is_sum = cmd == "summarize";
    for (int i = first; i < argc; i++) {
        const std::string s = argv[i];
        if (s == "-h" || s == "--help") { usage(stdout); return 0; }
        else if (!is_list && !is_sum && (s == "-m" || s == "--max-query-len")) { a.has_mql = true; a.mql = strtoull(need(i, "-m"), nullptr, 10); }
        else if (!is_sum && (s == "-o" || s == "--output")) a.output = need(i, "-o");
        else if (!is_list && !is_sum && (s == "-l" || s == "--low-memory")) {}            // access mode only: the file is mapped
        else if (!is_sum && (s == "-v" || s == "--very-low-memory")) {}
        else if (!is_list && !is_sum && s == "--device") a.device = atoi(need(i, "--device"));
        else if (is_locate && (s == "-a" || s == "--abs")) a.abs = true;
        else if (is_extract && (s == "-p" || s == "--prefix-len")) { a.has_prefix = true; a.prefix_len = strtoull(need(i, "-p"), nullptr, 10); }
        else if (is_extract && (s == "-s" || s == "--suffix-len")) { a.has_suffix = true; a.suffix_len = strtoull(need(i, "-s"), nullptr, 10); }
        else if (is_list && (s == "-r" || s == "--show-rank")) a.show_rank = true;
        else if (is_list && (s == "-s" || s == "--show-suffix")) a.show_suffix = true;
        else if (is_list && (s == "-p" || s == "--show-lcp")) a.show_lcp = true;
        else if (is_list && s == "--len") { a.has_len = true; a.len = strtoull(need(i, "--len"), nullptr, 10); }
        else if (is_list && (s == "-n" || s == "--number")) { a.has_number = true; a.number = strtoull(need(i, "-n"), nullptr, 10); }
        else if (is_list && s.size() > 2 && s[0] == '-' && s[1] != '-' && s.find_first_not_of("rspv", 1) == std::string::npos) {
            for (size_t k = 1; k < s.size(); k++) {     // combined short flags: -rsp
                if (s[k] == 'r') a.show_rank = true; else if (s[k] == 's') a.show_suffix = true; else if (s[k] == 'p') a.show_lcp = true;
            }
        }
        else if (!s.empty() && s[0] == '-' && s.size() > 1) { fprintf(stderr, "error: unexpected argument '%s'\n", s.c_str()); return 2; }
        else pos.push_back(s);
    }
    if (pos.empty()) { fprintf(stderr, "error: the following required arguments were not provided:\n  <%s>\n", is_list ? "FILE" : "SUFR"); return 2; }
    a.file = pos[0];
    a.positional.assign(pos.begin() + 1, pos.end());
    if (!is_list && !is_sum && a.positional.empty()) {
        fprintf(stderr, "error: the following required arguments were not provided:\n  <QUERY>...\n");
        return 2;
    }
    if (cmd == "count") return cmd_count(a);
    if (is_locate) return cmd_locate(a);
    if (is_extract) return cmd_extract(a);
    if (is_list) return cmd_list(a);
    return cmd_summarize(a);
}

}  // namespace

int main(int argc, char** argv)
{
    const auto t_main = std::chrono::steady_clock::now();
    const double main_epoch = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    Log log;
    std::string log_file, input, output, seed_mask, delim = "%";
    int device = 0;
    std::vector<int> devices;
    uint64_t window = 0, margin = 0, array_budget = 0;
    bool have_cmd = false, have_output = false, have_mask = false;
    int threads = 0;
    sufr_create_args a;
    memset(&a, 0, sizeof a);
    a.num_partitions = 16;
    a.random_seed = 42;

    auto need = [&](int& i, const char* opt) -> const char* {
        if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", opt); exit(2); }
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i];
        if (s == "-h" || s == "--help") return usage(stdout);
        else if (s == "-t" || s == "--threads") threads = atoi(need(i, "--threads"));
        else if (s == "-l" || s == "--log") {
            std::string v = need(i, "--log");
            if (v == "info") log.level = 1; else if (v == "debug") log.level = 2;
            else { fprintf(stderr, "error: invalid value '%s' for '--log <LOG>'\n", v.c_str()); return 2; }
        }
        else if (s == "--log-file") log_file = need(i, "--log-file");
        else if (s == "--device") device = atoi(need(i, "--device"));
        else if (s == "--devices") {
            std::string v = need(i, "--devices");
            devices.clear();
            for (size_t p = 0; p <= v.size();) {
                size_t q = v.find(',', p);
                if (q == std::string::npos) q = v.size();
                if (q == p || v.substr(p, q - p).find_first_not_of("0123456789") != std::string::npos) {
                    fprintf(stderr, "error: invalid value '%s' for '--devices <ID,ID,...>'\n", v.c_str());
                    return 2;
                }
                devices.push_back(atoi(v.substr(p, q - p).c_str()));
                p = q + 1;
            }
        }
        else if (!have_cmd && (s == "create" || s == "cr")) have_cmd = true;
        else if (!have_cmd && (s == "count" || s == "co")) return run_query("count", argc, argv, i + 1, threads);
        else if (!have_cmd && (s == "locate" || s == "lo")) return run_query("locate", argc, argv, i + 1, threads);
        else if (!have_cmd && (s == "extract" || s == "ex")) return run_query("extract", argc, argv, i + 1, threads);
        else if (!have_cmd && (s == "list" || s == "ls")) return run_query("list", argc, argv, i + 1, threads);
        else if (!have_cmd && (s == "summarize" || s == "su")) return run_query("summarize", argc, argv, i + 1, threads);
        else if (!have_cmd) { fprintf(stderr, "error: unrecognized subcommand '%s'\n", s.c_str()); return 2; }
        else if (s == "-n" || s == "--num-partitions") a.num_partitions = strtoull(need(i, "-n"), nullptr, 10);
        else if (s == "-m" || s == "--max-query-len") { a.has_max_query_len = 1; a.max_query_len = strtoull(need(i, "-m"), nullptr, 10); }
        else if (s == "-o" || s == "--output") { output = need(i, "-o"); have_output = true; }
        else if (s == "-d" || s == "--dna") a.is_dna = 1;
        else if (s == "-a" || s == "--allow-ambiguity") a.allow_ambiguity = 1;
        else if (s == "-i" || s == "--ignore-softmask") a.ignore_softmask = 1;
        else if (s == "-D" || s == "--sequence-delimiter") delim = need(i, "-D");
        else if (s == "-s" || s == "--seed-mask") { seed_mask = need(i, "-s"); have_mask = true; }
        else if (s == "-r" || s == "--random-seed") a.random_seed = strtoull(need(i, "-r"), nullptr, 10);
        else if (s == "--window") window = strtoull(need(i, "--window"), nullptr, 10);
        else if (s == "--array-budget") array_budget = strtoull(need(i, "--array-budget"), nullptr, 10);
        else if (s == "--margin") margin = strtoull(need(i, "--margin"), nullptr, 10);
        else if (!s.empty() && s[0] == '-' && s.size() > 1) { fprintf(stderr, "error: unexpected argument '%s'\n", s.c_str()); return 2; }
        else if (input.empty()) input = s;
        else { fprintf(stderr, "error: unexpected argument '%s'\n", s.c_str()); return 2; }
    }
    if (!have_cmd || input.empty()) return usage(stderr);
    if (a.has_max_query_len && have_mask) {
        fprintf(stderr, "error: the argument '--max-query-len <CONTEXT>' cannot be used with '--seed-mask <MASK>'\n");
        return 2;
    }
    if (delim.size() != 1) { fprintf(stderr, "error: invalid value '%s' for '--sequence-delimiter <DELIM>'\n", delim.c_str()); return 2; }
    if (!log_file.empty()) {
        log.out = fopen(log_file.c_str(), "w");
        if (!log.out) { fprintf(stderr, "Error: %s: cannot open log file\n", log_file.c_str()); return 1; }
    }
    a.input = input.c_str();
    a.output = have_output ? output.c_str() : nullptr;
    a.sequence_delimiter = (uint8_t)delim[0];
    a.seed_mask = have_mask ? seed_mask.c_str() : nullptr;

    // the device context (HIP initialisation, ~0.3 s) comes up while the sequence file is being read
    auto t0 = std::chrono::steady_clock::now();
    if (devices.empty()) devices.push_back(device);
    std::vector<sufr_hip_ctx*> ctxs(devices.size(), nullptr);
    sufr_hip_ctx* ctx = nullptr;
    std::string ctx_error;
    double ctx_up_s = 0.0;
    std::thread bring_up([&]() {
        for (size_t r = 0; r < devices.size(); r++) {
            ctxs[r] = sufr_hip_create(devices[r]);
            if (!ctxs[r]) {                                    // thread-local in the library: read it here
                ctx_error = sufr_hip_last_error(nullptr);
                for (size_t q = 0; q < r; q++) { sufr_hip_destroy(ctxs[q]); ctxs[q] = nullptr; }
                return;
            }
        }
        ctx = ctxs[0];
        ctx_up_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
    sufr_sequence_data sd;
    char rerr[512] = {0};
    const int read_rc = sufr_read_sequence_file(a.input, a.sequence_delimiter, &sd, rerr, sizeof rerr);
    const double read_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bring_up.join();
    const double ready_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count();
    const double ctx_s = ctx_up_s;
    if (!ctx) {
        if (read_rc == 0) sufr_sequence_data_free(&sd);
        fprintf(stderr, "Error: %s\n", ctx_error.c_str());
        return 1;
    }
    auto destroy_all = [&]() { for (auto* c : ctxs) if (c) sufr_hip_destroy(c); };
    if (read_rc != 0) { fprintf(stderr, "Error: %s\n", rerr); destroy_all(); return 1; }
    {
        std::string ids;
        for (size_t r = 0; r < devices.size(); r++) ids += (r ? "," : "") + std::to_string(devices[r]);
        log.info(std::string("Using HIP device") + (devices.size() > 1 ? "s " : " ") + ids);
    }
    if (window || margin) for (auto* c : ctxs) sufr_hip_set_window(c, window, margin);
    if (array_budget) sufr_hip_set_array_budget(ctxs[0], array_budget);
    char path[4096];
    std::vector<sufr_hip_stats> sts(devices.size());
    memset(sts.data(), 0, sts.size() * sizeof(sufr_hip_stats));
    int rc = sufr_hip_create_from_sequence_multi(ctxs.data(), (int)ctxs.size(), &sd, &a, path, sizeof path, sts.data());
    sufr_hip_stats st = sts[0];
    for (size_t r = 1; r < sts.size(); r++) {                  // totals over the shards; device times: the slowest
        st.num_suffixes += sts[r].num_suffixes;
        if (sts[r].ms_total > st.ms_total) st.ms_total = sts[r].ms_total;
        if (sts[r].num_levels > st.num_levels) st.num_levels = sts[r].num_levels;
    }
    st.host_read_s = (float)read_s;
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0) {
        fprintf(stderr, "Error: %s\n", sufr_hip_last_error(ctx));
        sufr_sequence_data_free(&sd);
        destroy_all();
        return 1;
    }
    log.info("Read input of len " + with_commas(st.text_len));
    char buf[512];
    snprintf(buf, sizeof buf, "Partitioned %s suffixes on %u-char prefixes (%u radix passes of %u bits) in %.3fms",
             with_commas(st.num_suffixes).c_str(), st.num_passes * (st.digit_bits / (st.bits_per_char ? st.bits_per_char : 1)),
             st.num_passes, st.digit_bits, st.ms_hist_text + st.ms_partition + st.ms_passes);
    log.info(buf);
    snprintf(buf, sizeof buf, "Sorted %s suffixes in %u level%s in %.3fms (device total %.3fms)",
             with_commas(st.num_suffixes).c_str(), st.num_levels, st.num_levels == 1 ? "" : "s",
             st.ms_finish + st.ms_deep, st.ms_total);
    log.info(buf);
    struct stat sb;
    uint64_t bytes = stat(path, &sb) == 0 ? (uint64_t)sb.st_size : 0;
    snprintf(buf, sizeof buf, "Wrote %s byte%s to '%s' in %.3fs", with_commas(bytes).c_str(), bytes == 1 ? "" : "s", path, secs);
    log.info(buf);
    // the four host phases of the run (they add up to the time since main() was entered; what a caller's clock sees beyond
    // them is process start -- loading the HIP runtime -- and exit)
    snprintf(buf, sizeof buf, "host phases: start-up + read %.3fs (read %.3fs, device contexts up %.3fs), H2D + build %.3fs, "
             "D2H + write %.3fs, since main() %.3fs", ready_s, st.host_read_s, ctx_s, st.host_build_s, st.host_write_s,
             std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count());
    log.debug(buf);
    {   // wall-clock stamps of main()'s entry and of the exit below: a caller's clock around the process tells start and exit apart
        const double now_epoch = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        snprintf(buf, sizeof buf, "epoch: main %.6f exit %.6f", main_epoch, now_epoch);
        log.debug(buf);
    }
    if (log.out != stdout) fclose(log.out);
    // The file is closed and renamed into place: nothing is left that a teardown could add.  Handing back 3 GB of host
    // text takes 0.15 s of a 2 s run (measured, profiles/r04_e2e_repeat.txt); the process image goes away as a whole.
    fflush(stdout); fflush(stderr);
    _exit(0);
}
