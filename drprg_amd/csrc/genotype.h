// genotype.h -- coverage vector -> pandora_genotyped.vcf (host side, O(#sites)).
//
// Restates the tail of `pandora map --genotype --local` (estimate_parameters, build_vcf,
// add_sample_covgs_to_vcf, SampleInfo likelihood/GT/GT_CONF).  The output surface is pinned by
// /root/reference/tests/cases/predict/ERR4796933.pandora.vcf and consumed by
// /root/reference/src/lib.rs:935-1181, src/filter.rs:50-55, src/minor.rs:85-97.
#pragma once
#include "index.h"
#include "params.h"

namespace drprg {

struct AlleleStats {
    uint32_t mean_fwd = 0, mean_rev = 0, med_fwd = 0, med_rev = 0, sum_fwd = 0, sum_rev = 0;
    double gaps = 1.0;
    double likelihood = 0.0;
};

struct VcfRecord {
    std::string chrom;
    uint32_t pos = 0; // 1-based
    std::string ref;
    std::vector<std::string> alts;
    std::string vc, graphtype;
    std::vector<AlleleStats> alleles; // ref first
    std::vector<std::vector<uint32_t>> allele_knodes; // per allele: the global k-mer nodes its statistics were taken over
    int gt = 0;
    double gt_conf = 0.0;
};

// A stretch of a locus' called consensus that the reads do not support: the mapping half of `pandora discover`
// (/root/reference/src/lib.rs:513-578) ends with these; pandora then assembles the reads over each region locally.
struct CandidateRegion {
    std::string chrom;
    uint32_t prg = 0;              // index of the locus in the PRG file
    uint32_t start = 0, end = 0;   // 0-based half-open on the consensus sequence, padding included
    uint32_t low_start = 0, low_end = 0; // the low-coverage bases themselves
    uint32_t max_covg = 0;         // largest per-base coverage inside [low_start, low_end)
    std::string seq;               // consensus[start, end)
    std::string left_anchor, right_anchor; // the anchor_len bases of the consensus before `start` / after `end` (empty at a locus end)
    std::string left_context, right_context; // up to 3 anchor_len bases before / after: fallback anchors for noisy reads (whole multiples of anchor_len)
};

// the called consensus of a locus as a walk through its local graph (what pandora's denovo_paths.txt lists as "nodes")
struct ConsensusNode {
    uint32_t id, start, end; // local node, its [start, end) in the PRG string
    std::string seq;
};
struct LocusConsensus {
    std::string chrom;
    uint32_t prg = 0;
    std::vector<ConsensusNode> nodes;
    std::string seq;
};

// pandora discover's candidate-region options [UPSTREAM-MEMORY: --min-candidate-covg 3, --min-candidate-len 1,
// --max-candidate-len 50, --pad 22, --merge 22]
struct DiscoverParams {
    uint32_t min_candidate_covg = 3, min_candidate_len = 1, max_candidate_len = 50, padding = 22, merge_dist = 22;
    uint32_t anchor_len = 15;     // exact-match anchors either side of a region (denovo.cpp)
    uint32_t min_support = 3;     // reads that must spell the novel allele ...
    double min_fraction = 0.5;    // ... and their share of the reads that span the region
    uint32_t max_len_change = 30; // longest insertion / deletion accepted between the anchors
};

struct GenotypeResult {
    uint32_t exp_depth_covg = 1;
    uint32_t min_kmer_covg = 0;
    CoverageModel model;              // what estimate_parameters made of the k-mer coverages (params.h)
    std::vector<std::string> present; // loci with a ##contig line, sorted
    std::vector<std::string> absent;
    std::vector<std::string> dropped_low_coverage; // loci with clusters whose best path is almost bare in a deep sample (among `absent`)
    std::vector<VcfRecord> records;   // sorted by (chrom, pos, ref, alts)
    std::vector<CandidateRegion> candidates; // low-coverage regions of the called consensus of every present locus
    std::vector<LocusConsensus> consensus;   // of the loci that have candidate regions
};

// per-allele statistics from the k-mer coverages of one allele (SampleInfo)
AlleleStats allele_stats(const std::vector<uint32_t>& fwd, const std::vector<uint32_t>& rev, uint32_t min_kmer_covg);
double allele_likelihood(double e, double c_a, double c_other, double eps, double gaps);
void genotype_site(std::vector<AlleleStats>& alleles, double e, double eps, int& gt, double& gt_conf);
uint32_t estimate_exp_depth_covg(const std::vector<uint32_t>& kmer_total_covg, uint32_t zero_thresh);

// covg: u32[2*total_knodes] ([2g] fwd, [2g+1] rev); prg_reads: clusters placed per PRG;
// total_bases: bases mapped (for the genome-size coverage estimate); vcf_refs: genes.fa or "".
GenotypeResult genotype(const PrgIndex& idx, const std::vector<uint32_t>& covg, const std::vector<uint32_t>& prg_reads,
    uint64_t total_bases, const MapParams& p, const std::string& vcf_refs, const DiscoverParams& dp = DiscoverParams());

// runs of bases whose coverage is <= min_covg (covered[i] == 0: no k-mer of the path reaches base i, never part of a run),
// kept when min_len <= length <= max_len; half-open intervals (pandora identify_low_coverage_intervals)
std::vector<std::pair<uint32_t, uint32_t>> low_coverage_intervals(const std::vector<uint32_t>& covg, const std::vector<uint8_t>& covered,
    uint32_t min_covg, uint32_t min_len, uint32_t max_len);

void write_vcf(const std::string& path, const GenotypeResult& r, const std::string& sample);
std::string format_g(double v); // ostream default formatting (%g, 6 significant digits)

} // namespace drprg
