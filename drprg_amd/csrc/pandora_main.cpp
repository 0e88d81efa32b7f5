// pandora_main.cpp -- drop-in executable for the `pandora` binary drprg shells out to.
//
// Accepts exactly the argv struct Pandora builds (/root/reference/src/lib.rs:479-642):
//   pandora index    -t T -w W -k K <prg>
//   pandora discover -g G --max-covg M -v -o <dir> -t T -w W -k K -c C [-I] [-K] <prg> <query.tsv>
//   pandora map      --genotype --local --gt-conf 0 -v -o <out> -g G --max-covg M --vcf-refs <genes.fa>
//                    -t T -w W -k K -c C [-I] [-K] <prg> <reads>
// and writes the files drprg then looks for: <prg>.k<K>.w<W>.idx + kmer_prgs/ (index),
// <dir>/denovo_paths.txt (discover, /root/reference/src/lib.rs:569-577),
// <out>/pandora_genotyped.vcf (map, /root/reference/src/lib.rs:644-646).
// Diagnostics go to stderr, progress to stdout (drprg redirects stdout to a log file); non-zero exit
// on failure (/root/reference/src/lib.rs:497-506).  Use it with `drprg predict -p <this file>`.
#include "../../include/drprg_hip.h"
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

// 2-bit packed ingest (include/drprg_hip.h "packed reads") unless DRPRG_HIP_INPUT=ascii: a quarter of the bytes to page-lock, to move over
// PCIe and to keep in HBM; same results
static int packed_input()
{
    const char* e = std::getenv("DRPRG_HIP_INPUT");
    return !(e && std::string(e) == "ascii");
}


struct Args {
    std::string cmd;
    int threads = 1, w = 14, k = 15;
    uint32_t min_cluster_size = 10;
    bool illumina = false, genotype = false, local = false, verbose = false, clean = false, binomial = false;
    double error_rate = -1, gt_conf = 1;
    int max_diff = -1;
    uint64_t genome_size = 5000000;
    uint64_t max_covg = 300;
    std::string outdir = "pandora", vcf_refs;
    std::vector<std::string> positional;
    int device = 0;
};

[[noreturn]] void die(const std::string& msg, int code = 1)
{
    std::fprintf(stderr, "pandora (drprg-hip): error: %s\n", msg.c_str());
    std::exit(code);
}

void usage()
{
    std::fprintf(stderr,
        "pandora-compatible front end of the MI355X drprg hot path\n"
        "  pandora index    [-t N] [-w W] [-k K] <prg>\n"
        "  pandora map      [--genotype] [--local] [--gt-conf X] [-v] [-o DIR] [-g SIZE] [--max-covg N]\n"
        "                   [--vcf-refs FASTA] [-t N] [-w W] [-k K] [-c N] [-I] [-K] [-e RATE] [--max-diff N] <prg> <reads>\n"
        "  pandora discover [same mapping options] <prg> <query.tsv>\n"
        "environment: DRPRG_HIP_DEVICE selects the GPU (default 0); DRPRG_HIP_DEVICES=0,1,.. maps on several GPUs of the node\n");
}

Args parse(int argc, char** argv)
{
    Args a;
    if (argc < 2) {
        usage();
        std::exit(2);
    }
    a.cmd = argv[1];
    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) die(std::string("option ") + argv[i] + " needs a value", 2);
        return argv[++i];
    };
    for (int i = 2; i < argc; ++i) {
        std::string s = argv[i];
        if (s == "-t" || s == "--threads") a.threads = std::atoi(need(i));
        else if (s == "-w") a.w = std::atoi(need(i));
        else if (s == "-k") a.k = std::atoi(need(i));
        else if (s == "-c" || s == "--min-cluster-size") a.min_cluster_size = (uint32_t)std::strtoul(need(i), nullptr, 10);
        else if (s == "-g" || s == "--genome-size") a.genome_size = std::strtoull(need(i), nullptr, 10);
        else if (s == "--max-covg") a.max_covg = std::strtoull(need(i), nullptr, 10);
        else if (s == "-o" || s == "--outdir") a.outdir = need(i);
        else if (s == "--vcf-refs") a.vcf_refs = need(i);
        else if (s == "--gt-conf") a.gt_conf = std::atof(need(i));
        else if (s == "-e" || s == "--error-rate") a.error_rate = std::atof(need(i));
        else if (s == "-m" || s == "--max-diff") a.max_diff = std::atoi(need(i));
        else if (s == "-I" || s == "--illumina") a.illumina = true;
        else if (s == "--bin") a.binomial = true;
        else if (s == "-K" || s == "--debugging-files") a.clean = false;
        else if (s == "--genotype") a.genotype = true;
        else if (s == "--local") a.local = true;
        else if (s == "-v") a.verbose = true;
        else if (s == "-h" || s == "--help") {
            usage();
            std::exit(0);
        } else if (!s.empty() && s[0] == '-' && s.size() > 1) die("unknown option " + s, 2);
        else a.positional.push_back(s);
    }
    if (const char* d = std::getenv("DRPRG_HIP_DEVICE")) a.device = std::atoi(d);
    return a;
}

void make_dirs(const std::string& path)
{
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) die("cannot create directory " + cur);
        }
        if (i < path.size()) cur += path[i];
    }
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

drprg_hip_ctx* open_ctx(const Args& a)
{
    // DRPRG_HIP_DEVICES=0,1,2,3: one context over several GPUs of the node (the reads shard by ingest block)
    std::vector<int> devices;
    if (const char* d = std::getenv("DRPRG_HIP_DEVICES"))
        for (const char* p = d; *p;) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int)v);
            p = *end ? end + 1 : end;
        }
    drprg_hip_ctx* ctx = devices.size() > 1 ? drprg_hip_open_multi(a.positional[0].c_str(), a.w, a.k, devices.data(), (int)devices.size(), 1)
                                            : drprg_hip_open(a.positional[0].c_str(), a.w, a.k, devices.size() == 1 ? devices[0] : a.device);
    if (!ctx) die(std::string("cannot open index for ") + a.positional[0] + ": " + drprg_hip_last_error(nullptr));
    drprg_hip_map_opts o {};
    o.illumina = a.illumina;
    o.binomial = a.binomial;
    o.error_rate = a.error_rate;
    o.max_diff = a.max_diff;
    o.min_cluster_size = a.min_cluster_size;
    o.genome_size = a.genome_size;
    if (int rc = drprg_hip_set_opts(ctx, &o)) die(drprg_hip_last_error(ctx), -rc);
    drprg_hip_set_threads(ctx, a.threads);
    drprg_hip_set_input_format(ctx, packed_input()); // the parser threads pack the reads to 2 bits (DRPRG_HIP_INPUT=ascii: one byte per base)
    return ctx;
}

int cmd_index(const Args& a)
{
    if (a.positional.size() != 1) die("index needs exactly one PRG file", 2);
    double t0 = now_s();
    int rc = drprg_hip_index(a.positional[0].c_str(), a.w, a.k, a.threads);
    if (rc) die(drprg_hip_last_error(nullptr), -rc);
    std::printf("[pandora-hip] indexed %s (w=%d k=%d) in %.2fs\n", a.positional[0].c_str(), a.w, a.k, now_s() - t0);
    return 0;
}

void report_counters(drprg_hip_ctx* ctx, double secs)
{
    uint64_t c[8];
    drprg_hip_counters(ctx, c);
    std::printf("[pandora-hip] reads=%llu bases=%llu minimizers=%llu hits=%llu clusters=%llu hits_in_clusters=%llu in %.3fs (%.0f reads/s)\n",
        (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[3],
        (unsigned long long)c[4], (unsigned long long)c[5], secs, secs > 0 ? (double)c[0] / secs : 0.0);
}

// identifies (PRG file, reads file, mapping parameters): what a cached coverage vector belongs to
std::string run_tag(const Args& a, const std::string& reads)
{
    auto stamp = [](const std::string& p) {
        struct stat st;
        char* rp = realpath(p.c_str(), nullptr);
        std::string s = rp ? rp : p;
        std::free(rp);
        if (stat(p.c_str(), &st) == 0)
            s += ":" + std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtim.tv_sec) + "."
                + std::to_string((long long)st.st_mtim.tv_nsec);
        return s;
    };
    return stamp(a.positional[0]) + "|" + stamp(reads) + "|w" + std::to_string(a.w) + "|k" + std::to_string(a.k) + "|c"
        + std::to_string(a.min_cluster_size) + "|I" + std::to_string((int)a.illumina) + "|e" + std::to_string(a.error_rate) + "|m"
        + std::to_string(a.max_diff) + "|g" + std::to_string((unsigned long long)a.genome_size);
}

const char* COVERAGE_CACHE = ".drprg_hip_coverage";

int cmd_map(const Args& a)
{
    if (a.positional.size() != 2) die("map needs <prg> <reads>", 2);
    make_dirs(a.outdir);
    drprg_hip_ctx* ctx = open_ctx(a);
    double t0 = now_s();
    // drprg runs `discover -o <out>/discover` on the same reads and PRG first (/root/reference/src/predict.rs:248-255): if that
    // pass left its coverage vector there, take it instead of streaming the reads again
    const std::string cache = a.outdir + "/discover/" + COVERAGE_CACHE;
    uint64_t cached[8];
    if (drprg_hip_load_coverage(ctx, cache.c_str(), run_tag(a, a.positional[1]).c_str(), cached) == 0) {
        std::printf("[pandora-hip] coverage of this PRG and these reads taken from %s (reads=%llu bases=%llu hits=%llu): no second mapping pass\n",
            cache.c_str(), (unsigned long long)cached[0], (unsigned long long)cached[1], (unsigned long long)cached[3]);
    } else {
        if (int rc = drprg_hip_map_fastx(ctx, a.positional[1].c_str())) die(drprg_hip_last_error(ctx), -rc);
        report_counters(ctx, now_s() - t0);
    }
    const std::string vcf = a.outdir + "/pandora_genotyped.vcf";
    if (int rc = drprg_hip_genotype(ctx, a.vcf_refs.empty() ? nullptr : a.vcf_refs.c_str(), vcf.c_str(), "sample"))
        die(drprg_hip_last_error(ctx), -rc);
    uint32_t gi[4];
    drprg_hip_genotype_info(ctx, gi);
    std::printf("[pandora-hip] exp_depth_covg=%u min_kmer_covg=%u loci_present=%u records=%u -> %s\n", gi[0], gi[1], gi[2], gi[3],
        vcf.c_str());
    drprg_hip_close(ctx);
    return 0;
}

int cmd_discover(const Args& a)
{
    if (a.positional.size() != 2) die("discover needs <prg> <query.tsv>", 2);
    make_dirs(a.outdir);
    // query.tsv: one line "sample<TAB>/abs/reads" (/root/reference/src/predict.rs:227-231)
    std::ifstream q(a.positional[1]);
    if (!q) die("cannot open " + a.positional[1]);
    std::string sample, reads;
    if (!(q >> sample >> reads)) die("malformed query file " + a.positional[1]);
    drprg_hip_ctx* ctx = open_ctx(a);
    // pandora discover walks the reads twice (mapping, then the candidate regions); here the second walk is over the blocks the
    // first one left in HBM (up to DRPRG_HIP_KEEP_READS_GB per device, default 32, 0 = read the file again)
    {
        double keep_gb = 32;
        if (const char* e = std::getenv("DRPRG_HIP_KEEP_READS_GB")) keep_gb = std::atof(e);
        if (keep_gb > 0)
            if (int rc = drprg_hip_keep_reads(ctx, (uint64_t)(keep_gb * 1e9))) die(drprg_hip_last_error(ctx), -rc);
    }
    double t0 = now_s();
    if (int rc = drprg_hip_map_fastx(ctx, reads.c_str())) die(drprg_hip_last_error(ctx), -rc);
    report_counters(ctx, now_s() - t0);
    // The mapping half of discover is the same kernels as `map`; its products are (1) the candidate regions -- stretches of each
    // locus' called consensus that the reads do not support --, (2) the novel variants a host-side pile-up of
    // the reads finds in them, and (3) the coverage vector, kept for the `map` call drprg issues next on the unchanged PRG.
    // denovo_paths.txt lists the loci with novel variants (the caller then runs `make_prg update` on this file,
    // /root/reference/src/lib.rs:279-456): the layout of the example in the reference tree (/root/reference/src/lib.rs:3010-3038),
    // read back in the tests by a line-for-line port of the reference's own parser (/root/reference/src/lib.rs:648-697) for files
    // with several loci, several variants per locus and paths through nested sites.  DRPRG_HIP_DENOVO_PATHS=0: report 0 loci
    // (drprg then keeps the index PRG, /root/reference/src/lib.rs:299-301); the findings are in denovo_variants.tsv either way.
    const char* paths_env = std::getenv("DRPRG_HIP_DENOVO_PATHS");
    const int list_loci = !(paths_env && std::atoi(paths_env) == 0);
    uint32_t found[3] = { 0, 0, 0 };
    if (int rc = drprg_hip_discover_reads(ctx, reads.c_str(), nullptr, a.outdir.c_str(), sample.c_str(), list_loci, found))
        die(drprg_hip_last_error(ctx), -rc);
    if (drprg_hip_save_coverage(ctx, (a.outdir + "/" + COVERAGE_CACHE).c_str(), run_tag(a, reads).c_str()) != 0)
        std::fprintf(stderr, "pandora (drprg-hip): warning: could not keep the coverage vector for `map`: %s\n", drprg_hip_last_error(ctx));
    if (found[1] && !list_loci)
        std::fprintf(stderr,
            "pandora (drprg-hip): WARNING: %u novel variant(s) in %u locus/loci found (%s/denovo_variants.tsv) but denovo_paths.txt reports 0 "
            "loci (DRPRG_HIP_DENOVO_PATHS=0), so the PRG will not be updated.\n", found[1], found[2], a.outdir.c_str());
    uint64_t ri[4] = { 0, 0, 0, 0 };
    drprg_hip_resident_info(ctx, ri);
    std::printf("[pandora-hip] discover: %u candidate regions, %u novel variants in %u loci%s; reads of the second pass %s\n", found[0], found[1], found[2],
        list_loci ? "" : " (not listed in denovo_paths.txt)", ri[3] ? "resident in device memory" : "from the file");
    drprg_hip_close(ctx);
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    Args a = parse(argc, argv);
    if (a.cmd == "index") return cmd_index(a);
    if (a.cmd == "map") return cmd_map(a);
    if (a.cmd == "discover") return cmd_discover(a);
    if (a.cmd == "-h" || a.cmd == "--help") {
        usage();
        return 0;
    }
    usage();
    return 2;
}
