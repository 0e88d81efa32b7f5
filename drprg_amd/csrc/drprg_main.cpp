// drprg_main.cpp -- `drprg predict` front end on top of the C ABI (include/drprg_hip.h).
//
// Mirrors the CLI surface of /root/reference/src/predict.rs:134-202 (+ Filterer src/filter.rs:165-197, MinorAllele
// src/minor.rs:19-49, global -v/-t src/cli.rs:81-93) and the sequence of Predict::run (src/predict.rs:204-317):
// validate index -> map on the GPU -> discover from that same pass (candidate regions; novel variants from a pile-up of the reads;
// PRG update + index + second mapping pass only if there are any) -> genotype
// -> pandora_genotyped.vcf -> <sample>.drprg.vcf -> <sample>.drprg.json.  Host orchestration is C++ here because
// the image has no Rust toolchain; a Rust drprg binds the same ABI.
#include "../../include/drprg_hip.h"
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

[[noreturn]] void die(const std::string& m, int code = 1)
{
    std::fprintf(stderr, "drprg (hip): error: %s\n", m.c_str());
    std::exit(code);
}
bool exists(const std::string& p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}
bool is_dir(const std::string& p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

// -x <path | species[@version]> (/root/reference/src/cli.rs:21-78); named indexes live in ~/.drprg/<species>/<species>-<version>
// *named: the index was found by species name, i.e. it is one `drprg index --download` fetched (/root/reference/src/index.rs:122):
// its k-mer graphs and .idx are the real pandora's files
std::string resolve_index(const std::string& s, bool* named = nullptr)
{
    if (named) *named = false;
    if (exists(s)) return s;
    if (named) *named = true;
    if (s.find('/') != std::string::npos) die("Received an index which is path-like but does not exist");
    std::string species = s, version = "latest";
    size_t at = s.find('@');
    if (at != std::string::npos) {
        species = s.substr(0, at);
        version = s.substr(at + 1);
    }
    const char* home = std::getenv("HOME");
    std::string base = std::string(home ? home : ".") + "/.drprg/" + species;
    if (!is_dir(base)) die("No index for species " + species + " found in " + std::string(home ? home : ".") + "/.drprg");
    if (version != "latest") {
        std::string p = base + "/" + species + "-" + version;
        if (!exists(p)) die("Version " + version + " does not exist for species " + species);
        return p;
    }
    std::vector<std::string> dirs;
    if (DIR* d = opendir(base.c_str())) {
        while (dirent* e = readdir(d))
            if (e->d_name[0] != '.' && is_dir(base + "/" + e->d_name)) dirs.push_back(base + "/" + e->d_name);
        closedir(d);
    }
    if (dirs.empty()) die("No index versions found in " + species + " directory " + base);
    std::string best = dirs[0];
    for (auto& d : dirs)
        if (d > best) best = d;
    return best;
}

// first dr.prg.k<K>.w<W>.idx in the index directory (find_prg_index_in, /root/reference/src/lib.rs:1222-1231)
bool find_prg_index(const std::string& dir, int& k, int& w)
{
    bool found = false;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            int kk, ww;
            char tail[8] = { 0 };
            if (std::sscanf(e->d_name, "dr.prg.k%d.w%d.%3s", &kk, &ww, tail) == 3 && std::strcmp(tail, "idx") == 0) {
                k = kk;
                w = ww;
                found = true;
                break;
            }
        }
        closedir(d);
    }
    return found;
}

std::string file_prefix(const std::string& path) // PathExt::prefix: file name up to the first '.'
{
    size_t s = path.find_last_of('/');
    std::string name = s == std::string::npos ? path : path.substr(s + 1);
    size_t dot = name.find('.', name.empty() || name[0] != '.' ? 0 : 1);
    return dot == std::string::npos ? name : name.substr(0, dot);
}

void usage()
{
    std::fprintf(stderr,
        "drprg predict -x <index dir | species[@version]> -i <reads.fq[.gz]> [-o DIR] [-s SAMPLE] [-I] [-S]\n"
        "              [-f MAF] [-d MIN_COVG] [-D MAX_COVG] [-b MIN_STRAND_BIAS] [-g MIN_GT_CONF] [-L MAX_INDEL] [-K MIN_FRS]\n"
        "              [-C MIN_CLUSTER_SIZE] [--debug] [-v] [-t THREADS] [--rebuild-index]\n"
        "MI355X-native hot path; -p/-m/-M (external tools) are accepted and not needed: novel variants update the PRG in process.\n");
}

} // namespace

// seconds since main() was entered, for the -v lines (where the wall time of a prediction goes); since the moment the parent
// process names in DRPRG_HIP_T0 (seconds since the epoch, e.g. `DRPRG_HIP_T0=$(date +%s.%N)`), when it does -- that adds the
// time the loader takes before main()
static double since_start()
{
    static const double t0 = [] {
        const char* e = std::getenv("DRPRG_HIP_T0");
        const double given = e ? std::atof(e) : 0;
        return given > 0 ? given : std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    }();
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() - t0;
}

// 2-bit packed ingest (include/drprg_hip.h "packed reads") unless DRPRG_HIP_INPUT=ascii: a quarter of the bytes to page-lock, to move over
// PCIe and to keep in HBM; same results
static int packed_input()
{
    const char* e = std::getenv("DRPRG_HIP_INPUT");
    return !(e && std::string(e) == "ascii");
}

int main(int argc, char** argv)
{
    const double at_main = since_start();
    if (argc < 2 || std::strcmp(argv[1], "predict") != 0) {
        usage();
        return argc >= 2 && (!std::strcmp(argv[1], "-h") || !std::strcmp(argv[1], "--help")) ? 0 : 2;
    }
    std::string index, input, outdir = ".", sample;
    bool illumina = false, verbose = false, maf_given = false, rebuild_index = false;
    int threads = 1;
    uint32_t min_cluster = 10;
    drprg_hip_annotate_opts ao {};
    ao.min_covg = 3;
    ao.max_covg = INT_MAX;
    ao.min_strand_bias = 0.01f;
    ao.min_gt_conf = 0.0f;
    ao.min_frs = 0.0f;
    ao.max_indel = -1;
    ao.maf = 1.0f;
    ao.max_gaps = 0.5f;
    ao.max_called_gaps = 0.39f;
    ao.max_gaps_diff = 0.2f;
    ao.minor_min_covg = 3;
    ao.minor_min_strand_bias = 0.01f;
    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) die(std::string("option ") + argv[i] + " needs a value", 2);
        return argv[++i];
    };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-x" || a == "--index") index = need(i);
        else if (a == "-i" || a == "--input") input = need(i);
        else if (a == "-o" || a == "--outdir") outdir = need(i);
        else if (a == "-s" || a == "--sample") sample = need(i);
        else if (a == "-I" || a == "--illumina") illumina = true;
        else if (a == "-S" || a == "--ignore-synonymous") ao.ignore_synonymous = 1;
        else if (a == "-f" || a == "--maf") { ao.maf = std::strtof(need(i), nullptr); maf_given = true; }
        else if (a == "--max-gaps") ao.max_gaps = std::strtof(need(i), nullptr);
        else if (a == "--max-called-gaps") ao.max_called_gaps = std::strtof(need(i), nullptr);
        else if (a == "--max-gaps-diff") ao.max_gaps_diff = std::strtof(need(i), nullptr);
        else if (a == "--minor-min-covg") ao.minor_min_covg = std::atoi(need(i));
        else if (a == "--minor-min-strand-bias") ao.minor_min_strand_bias = std::strtof(need(i), nullptr);
        else if (a == "-d" || a == "--min-covg") ao.min_covg = std::atoi(need(i));
        else if (a == "-D" || a == "--max-covg") ao.max_covg = std::atoi(need(i));
        else if (a == "-b" || a == "--min-strand-bias") ao.min_strand_bias = std::strtof(need(i), nullptr);
        else if (a == "-g" || a == "--min-gt-conf") ao.min_gt_conf = std::strtof(need(i), nullptr);
        else if (a == "-L" || a == "--max-indel") ao.max_indel = std::atoi(need(i));
        else if (a == "-K" || a == "--min-frs") ao.min_frs = std::strtof(need(i), nullptr);
        else if (a == "-C" || a == "--pandora-min-cluster-size") min_cluster = (uint32_t)std::atoi(need(i));
        else if (a == "-p" || a == "--pandora" || a == "-m" || a == "--makeprg" || a == "-M" || a == "--mafft") (void)need(i);
        else if (a == "-t" || a == "--threads") threads = std::atoi(need(i));
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "--debug") verbose = true;
        else if (a == "--rebuild-index") rebuild_index = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else die("unknown option " + a, 2);
    }
    if (index.empty() || input.empty()) {
        usage();
        return 2;
    }
    if (illumina && !maf_given) ao.maf = 0.1f; // default_value_if("is_illumina", .., "0.1"), src/minor.rs:26-33
    if (!exists(input)) die(input + " does not exist");
    bool named_index = false;
    index = resolve_index(index, &named_index);
    // An index found by species name is a downloaded one: its dr.prg.k*.w*.idx and kmer_prgs/ are the real pandora's files.  If
    // they do not parse in the layout this build reads, rebuilding the graphs from dr.prg would run -- and silently hide that the
    // two programs may not agree on the k-mer graphs the coverage is counted on -- so that is an error here unless the user asks
    // for the rebuild (--rebuild-index / DRPRG_HIP_REBUILD_INDEX=1).  An index given as a path keeps the rebuild with a warning.
    {
        const char* rb = std::getenv("DRPRG_HIP_REBUILD_INDEX");
        if (named_index && !rebuild_index && !(rb && *rb && *rb != '0')) setenv("DRPRG_HIP_STRICT_INDEX", "1", 1);
    }
    if (mkdir(outdir.c_str(), 0777) != 0 && errno != EEXIST) die("Failed to create output directory " + outdir);
    // validate_index (/root/reference/src/predict.rs:400-418)
    int k = 0, w = 0;
    if (!find_prg_index(index, k, w)) die("Index is not valid due to missing file " + index + "/dr.prg.kX.wY.idx");
    for (const char* f : { "/.config.toml", "/dr.prg", "/kmer_prgs", "/panel.bcf", "/panel.bcf.csi", "/genes.fa", "/msas" })
        if (!exists(index + f)) die("Index is not valid due to missing file " + index + f);
    if (sample.empty()) sample = file_prefix(input);
    int device = 0;
    if (const char* d = std::getenv("DRPRG_HIP_DEVICE")) device = std::atoi(d);

    const std::string prg = index + "/dr.prg";
    // DRPRG_HIP_DEVICES=0,1,2,3: one context over several GPUs of the node (the reads shard by ingest block, the coverage
    // vectors are summed before the genotyping / report stages, which run once)
    std::vector<int> devices;
    if (const char* d = std::getenv("DRPRG_HIP_DEVICES"))
        for (const char* p = d; *p;) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int)v);
            p = *end ? end + 1 : end;
        }
    if (devices.size() == 1) device = devices[0];
    drprg_hip_ctx* ctx = devices.size() > 1 ? drprg_hip_open_multi(prg.c_str(), w, k, devices.data(), (int)devices.size(), 1)
                                            : drprg_hip_open(prg.c_str(), w, k, device);
    if (!ctx)
        die(std::string("cannot open the index: ") + drprg_hip_last_error(nullptr)
            + (named_index ? " (a downloaded index whose pandora files this build cannot read: --rebuild-index rebuilds the k-mer graphs from dr.prg)" : ""));
    drprg_hip_map_opts mo {};
    mo.illumina = illumina;
    mo.min_cluster_size = min_cluster;
    mo.genome_size = 4411532; // MTB_GENOME_SIZE, /root/reference/src/lib.rs:36
    if (int rc = drprg_hip_set_opts(ctx, &mo)) die(drprg_hip_last_error(ctx), -rc);
    drprg_hip_set_threads(ctx, threads);
    drprg_hip_set_input_format(ctx, packed_input()); // the parser threads pack the reads to 2 bits (DRPRG_HIP_INPUT=ascii: one byte per base)
    // The reads stay in HBM after the mapping pass (up to DRPRG_HIP_KEEP_READS_GB per device, default 32, 0 = off): discover takes
    // the few reads it needs from there and a novel variant maps them again from there -- the file is read once.
    double keep_gb = 32;
    if (const char* e = std::getenv("DRPRG_HIP_KEEP_READS_GB")) keep_gb = std::atof(e);
    if (keep_gb > 0)
        if (int rc = drprg_hip_keep_reads(ctx, (uint64_t)(keep_gb * 1e9))) die(drprg_hip_last_error(ctx), -rc);
    if (verbose && std::getenv("DRPRG_HIP_T0")) std::fprintf(stderr, "[drprg-hip +%.3fs] main() entered\n", at_main);
    if (verbose) std::fprintf(stderr, "[drprg-hip +%.3fs] mapping %s against %s (k=%d w=%d) on device %d\n", since_start(), input.c_str(), index.c_str(), k, w, device);
    // discover + map share ONE pass over the reads (the reference runs two, /root/reference/src/predict.rs:248-302)
    if (int rc = drprg_hip_map_fastx(ctx, input.c_str())) die(drprg_hip_last_error(ctx), -rc);
    if (verbose) std::fprintf(stderr, "[drprg-hip +%.3fs] reads mapped\n", since_start());
    {
        // discover (/root/reference/src/predict.rs:247-256): candidate regions of every locus' called consensus, then a host-side
        // pile-up of the reads over them (whole strings with -I, column-wise majority of aligned strings without).  Novel variants update the PRG (what MakePrg::update does with
        // make_prg + mafft in the reference, src/predict.rs:260-284, here a new site per variant: -m/-M are not needed), the updated
        // PRG is indexed in the output directory and the reads are mapped again against it.
        std::string ddir = outdir + "/discover";
        mkdir(ddir.c_str(), 0777);
        uint32_t found[3] = { 0, 0, 0 };
        if (int rc = drprg_hip_discover_reads(ctx, input.c_str(), (index + "/genes.fa").c_str(), ddir.c_str(), sample.c_str(), 1, found))
            die(drprg_hip_last_error(ctx), -rc);
        if (verbose || found[1]) {
            uint64_t ri[4] = { 0, 0, 0, 0 };
            drprg_hip_resident_info(ctx, ri);
            std::fprintf(stderr, "[drprg-hip +%.3fs] discover: %u candidate region(s), %u novel variant(s) in %u locus/loci (reads %s)\n", since_start(), found[0],
                found[1], found[2], ri[3] ? "resident in device memory" : "from the file");
        }
        if (found[1]) {
            const std::string updated = outdir + "/updated.dr.prg";
            uint32_t applied = 0;
            if (int rc = drprg_hip_update_prg(ctx, updated.c_str(), &applied)) die(drprg_hip_last_error(ctx), -rc);
            if (applied) {
                if (int rc = drprg_hip_index(updated.c_str(), w, k, threads)) die(drprg_hip_last_error(nullptr), -rc);
                // (the new context opens BEFORE the old one closes: the page-locked ingest blocks of the process go back to the
                // driver with its last context, and pinning them again costs more than the second index does in HBM)
                drprg_hip_ctx* next = devices.size() > 1 ? drprg_hip_open_multi(updated.c_str(), w, k, devices.data(), (int)devices.size(), 1)
                                                         : drprg_hip_open(updated.c_str(), w, k, device);
                if (!next) die(std::string("cannot open the updated PRG: ") + drprg_hip_last_error(nullptr));
                if (int rc = drprg_hip_set_opts(next, &mo)) die(drprg_hip_last_error(next), -rc);
                drprg_hip_set_threads(next, threads);
                drprg_hip_set_input_format(next, packed_input()); // the parser threads pack the reads to 2 bits (DRPRG_HIP_INPUT=ascii: one byte per base)
                // the reads again, against the updated index: from HBM if the first context kept them all, else from the file
                const int from_hbm = drprg_hip_map_resident(next, ctx);
                if (from_hbm != 0 && from_hbm != -61 /* ENODATA */) die(drprg_hip_last_error(next), -from_hbm);
                drprg_hip_close(ctx);
                ctx = next;
                if (from_hbm != 0)
                    if (int rc = drprg_hip_map_fastx(ctx, input.c_str())) die(drprg_hip_last_error(ctx), -rc);
                if (verbose)
                    std::fprintf(stderr, "[drprg-hip +%.3fs] %u novel site(s) added to %s; reads mapped again (%s)\n", since_start(), applied, updated.c_str(),
                        from_hbm == 0 ? "resident in device memory" : "from the file");
            }
        }
    }
    const std::string pandora_vcf = outdir + "/pandora_genotyped.vcf";
    if (int rc = drprg_hip_genotype(ctx, (index + "/genes.fa").c_str(), pandora_vcf.c_str(), "sample")) die(drprg_hip_last_error(ctx), -rc);
    if (verbose) {
        uint64_t c[8];
        uint32_t gi[4];
        drprg_hip_counters(ctx, c);
        drprg_hip_genotype_info(ctx, gi);
        std::fprintf(stderr, "[drprg-hip +%.3fs] reads=%llu hits=%llu clusters=%llu exp_depth_covg=%u loci_present=%u records=%u\n",
            since_start(), (unsigned long long)c[0], (unsigned long long)c[3], (unsigned long long)c[4], gi[0], gi[2], gi[3]);
    }
    drprg_hip_close(ctx);
    if (verbose) std::fprintf(stderr, "[drprg-hip +%.3fs] device context closed\n", since_start());
    char err[1024] = { 0 };
    const std::string out_vcf = outdir + "/" + sample + ".drprg.vcf", out_json = outdir + "/" + sample + ".drprg.json";
    if (int rc = drprg_hip_annotate(index.c_str(), pandora_vcf.c_str(), out_vcf.c_str(), &ao, err, sizeof err)) die(err, -rc);
    if (verbose) std::fprintf(stderr, "[drprg-hip +%.3fs] panel annotated: %s\n", since_start(), out_vcf.c_str());
    // <sample>.drprg.bcf: the file the reference leaves (/root/reference/src/predict.rs:429-431); the text VCF stays beside it
    if (int rc = drprg_hip_vcf_to_bcf(out_vcf.c_str(), (outdir + "/" + sample + ".drprg.bcf").c_str(), err, sizeof err)) die(err, -rc);
    if (int rc = drprg_hip_report_json(index.c_str(), out_vcf.c_str(), out_json.c_str(), sample.c_str(), -1, nullptr, err, sizeof err)) die(err, -rc);
    if (verbose) std::fprintf(stderr, "[drprg-hip +%.3fs] wrote %s\n", since_start(), out_json.c_str());
    // Every output file is closed.  Leave without the static destructors of the HIP runtime: measured (9 runs each, one box)
    // 0.15 s between the line above and the parent seeing the exit with `return 0`, 0.001 s this way for a sample without a
    // novel variant (with one, the kernel still takes ~0.13 s to take the process apart; hipDeviceReset first changes nothing).
    std::fflush(nullptr);
    // Not when anything is evidently attached that does its work from exit handlers (a profiler's output files, a sanitizer's
    // report, a preloaded library): then the process returns normally.  DRPRG_HIP_SLOW_EXIT=1 forces the normal return,
    // DRPRG_HIP_FAST_EXIT=0 as well.
    auto set = [](const char* name) { const char* e = std::getenv(name); return e && *e; };
    auto on = [](const char* name) { const char* e = std::getenv(name); return e && *e && *e != '0'; };
    bool tool = set("LD_PRELOAD") || set("ROCP_TOOL_LIBRARIES") || set("ROCPROFILER_REGISTER_FORCE_LOAD") || set("ROCPROF_OUTPUT_PATH")
        || set("ASAN_OPTIONS") || set("UBSAN_OPTIONS") || set("LSAN_OPTIONS") || set("GCOV_PREFIX") || set("LLVM_PROFILE_FILE");
#if defined(__SANITIZE_ADDRESS__)
    tool = true;
#endif
    if (const char* e = std::getenv("DRPRG_HIP_FAST_EXIT"); e && *e == '0') tool = true;
    if (tool || on("DRPRG_HIP_SLOW_EXIT")) return 0;
    _exit(0);
}
