// params.h -- what `pandora map` does between its read loop and the VCF: the coverage model of the sample (estimate_parameters),
// the maximum-likelihood path of every locus through its k-mer graph (KmerGraphWithCoverage::find_max_path) and the rule by
// which a locus whose best path has almost no coverage is dropped (LocalPRG::add_consensus_path_to_fastaq).  The reference
// reaches all of it through /root/reference/src/lib.rs:580-642 (`pandora map --genotype --local`) and consumes the result as
//   * e = exp_depth_covg in every LIKELIHOOD / GT_CONF of the VCF (filters: /root/reference/src/filter.rs:12-16, :149),
//   * the set of ##contig lines: a locus without one is reported absent (/root/reference/src/predict.rs:757-765).
// pandora's source is not in the reference tree: everything here is [UPSTREAM-MEMORY] (pandora 0.9 / 0.10: src/estimate_parameters.cpp,
// src/kmergraphwithcoverage.cpp, src/localPRG.cpp, src/utils.cpp); DESIGN.md section 4 lists every constant.  The test suite's checker
// states the same rules a second time, separately written, and tests/test_params.py compares the two.
#pragma once
#include "index.h"

namespace drprg {

struct CoverageModel {
    uint32_t exp_depth_covg = 1;
    bool bin = false;       // the binomial model is in force (asked for with --bin, or chosen because variance ~ mean)
    double e_rate = 0.11;   // k-mer error rate: -e, re-estimated in the binomial branch
    float nb_p = 0.015f, nb_r = 2.0f; // negative binomial parameters (pandora's defaults until fitted)
    double bin_p = 1.0;     // binomial model: probability that a read covering a k-mer shows it without error = 1 / exp(e_rate * k)
    int thresh = -25;       // log-probability threshold between "true" and "error" k-mers (find_max_path: paths that end at the sink)
    uint32_t num_reads = 0; // clusters per locus, averaged over the loci that have any (integer division)
    double mean = 0, var = 0; // of the k-mer coverage histogram at or above the zero threshold in force
    int branch = 0;         // 1 binomial, 2 negative binomial, 3 "insufficient coverage"
};

// histogram entries: fwd + rev coverage of every k-mer (not source / sink) of every locus with at least one cluster, values
// >= 1000 ignored.  global_covg = bases mapped / genome size (integer).
CoverageModel estimate_parameters(const std::vector<uint32_t>& kmer_covg, uint64_t clusters, uint64_t loci_with_clusters, uint32_t global_covg,
    int k, double e_rate, bool bin);

// log probability of a k-mer's coverage under the model (source and sink: 0)
float kmer_log_prob(const CoverageModel& m, uint32_t fwd, uint32_t rev, uint32_t num_reads_of_locus);
// thresh of the model from the log probabilities of all those k-mers (estimate_parameters' tail: find_prob_thresh)
int prob_threshold(const std::vector<float>& log_probs);

// maximum-likelihood path: k-mer node ids from the first real node to the last (source and sink excluded), maximising the mean
// log probability of the (at most max_kmers_to_average) next nodes.  logp[i] = kmer_log_prob of node i.
std::vector<uint32_t> find_max_path(const KmerGraph& kg, const std::vector<float>& logp, int thresh, uint32_t max_kmers_to_average = 100);

// per-base coverage (largest fwd + rev coverage of the path's k-mers that cover the base) of the local nodes the k-mers of `path`
// run through, in order
std::vector<uint32_t> base_coverage_along_path(const LocalGraph& g, const KmerGraph& kg, const std::vector<uint32_t>& path, const uint32_t* covg /* 2 per node */);
uint32_t mode_u32(std::vector<uint32_t> v); // pandora utils.cpp mode(): the smallest value that occurs most often, and at least twice; else 0
// a locus is dropped when the sample is deep (global_covg > 20) and both mode and mean of that coverage are below 3
bool path_coverage_too_low(const std::vector<uint32_t>& base_covg, uint32_t global_covg);

} // namespace drprg
