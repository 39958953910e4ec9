// sufr_cli.cpp -- `sufr create` on the MI355X build path.
//
// Command-line contract of the reference for this path: sufr/src/lib.rs:29-46 (global -t/--threads,
// -l/--log, --log-file), 83-125 (CreateArgs, alias `cr`), sufr/src/main.rs:8-40 (errors are printed as
// "Error: <msg>" and exit code 1).  Query sub-commands (count/extract/list/locate/summarize) are not
// part of the construction path and are not provided here.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sufr_hip.h"

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
            "Usage: sufr [OPTIONS] create|cr [OPTIONS] <INPUT>\n\n"
            "Global options:\n"
            "  -t, --threads <THREADS>   Accepted for compatibility (the build runs on the GPU)\n"
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
            "  -r, --random-seed <RANDSEED>      Random seed [default: 42]\n\n"
            "Texts of 2^32 - 2^24 bytes and more are refused (64-bit device indices are not built yet).\n");
    return f == stderr ? 2 : 0;
}

}  // namespace

int main(int argc, char** argv)
{
    Log log;
    std::string log_file, input, output, seed_mask, delim = "%";
    int device = 0;
    std::vector<int> devices;
    bool have_cmd = false, have_output = false, have_mask = false;
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
        else if (s == "-t" || s == "--threads") (void)need(i, "--threads");
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
        else if (!have_cmd) { fprintf(stderr, "error: unrecognized subcommand '%s' (this build provides `create`)\n", s.c_str()); return 2; }
        else if (s == "-n" || s == "--num-partitions") a.num_partitions = strtoull(need(i, "-n"), nullptr, 10);
        else if (s == "-m" || s == "--max-query-len") { a.has_max_query_len = 1; a.max_query_len = strtoull(need(i, "-m"), nullptr, 10); }
        else if (s == "-o" || s == "--output") { output = need(i, "-o"); have_output = true; }
        else if (s == "-d" || s == "--dna") a.is_dna = 1;
        else if (s == "-a" || s == "--allow-ambiguity") a.allow_ambiguity = 1;
        else if (s == "-i" || s == "--ignore-softmask") a.ignore_softmask = 1;
        else if (s == "-D" || s == "--sequence-delimiter") delim = need(i, "-D");
        else if (s == "-s" || s == "--seed-mask") { seed_mask = need(i, "-s"); have_mask = true; }
        else if (s == "-r" || s == "--random-seed") a.random_seed = strtoull(need(i, "-r"), nullptr, 10);
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
    });
    sufr_sequence_data sd;
    char rerr[512] = {0};
    const int read_rc = sufr_read_sequence_file(a.input, a.sequence_delimiter, &sd, rerr, sizeof rerr);
    const double read_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bring_up.join();
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
    sufr_sequence_data_free(&sd);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0) {
        fprintf(stderr, "Error: %s\n", sufr_hip_last_error(ctx));
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
    snprintf(buf, sizeof buf, "host phases: read %.3fs, H2D + build %.3fs, D2H + write %.3fs", st.host_read_s,
             st.host_build_s, st.host_write_s);
    log.debug(buf);
    destroy_all();
    if (log.out != stdout) fclose(log.out);
    return 0;
}
