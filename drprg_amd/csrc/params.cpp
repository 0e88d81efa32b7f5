// params.cpp -- see params.h.  Every rule below is a restatement of upstream pandora from memory [UPSTREAM-MEMORY]; nothing in
// /root/reference pins it except that the e it yields must be a positive integer (the seven fixture VCFs: SURVEY.md section 8a).
#include "params.h"
#include <algorithm>
#include <cmath>
#include <limits>

namespace drprg {

namespace {

constexpr size_t COVG_HIST = 1000; // pandora: kmer_covg_dist(1000, 0)
constexpr int PROB_HIST = 200;     // pandora: kmer_prob_dist(200, 0), log probabilities in (-200, 0]

// estimate_parameters.cpp fit_mean_covg / fit_variance_covg: moments of the histogram from `zero_thresh` upwards.  The threshold
// parameter is a uint8_t upstream, so a global coverage of 2560x and more wraps around; kept.
double fit_mean_covg(const std::vector<uint32_t>& hist, uint8_t zero_thresh)
{
    double sum = 0, total = 0;
    for (size_t i = zero_thresh; i < hist.size(); ++i) {
        sum += (double)hist[i] * (double)i;
        total += (double)hist[i];
    }
    return total == 0 ? 0.0 : sum / total;
}

double fit_variance_covg(const std::vector<uint32_t>& hist, double mean, uint8_t zero_thresh)
{
    double acc = 0, total = 0;
    for (size_t i = zero_thresh; i < hist.size(); ++i) {
        acc += ((double)i - mean) * ((double)i - mean) * (double)hist[i];
        total += (double)hist[i];
    }
    return total == 0 ? 0.0 : acc / total;
}

// find_mean_covg: the position of the highest point of the histogram behind its first (error) peak -- the walk ignores the
// falling flank, believes it has left the first peak after the fourth rise, and from then on keeps the largest bin
uint32_t find_mean_covg(const std::vector<uint32_t>& hist)
{
    bool first_peak = true;
    uint32_t max_covg = 0, noise_buffer = 0;
    for (uint32_t i = 1; i < hist.size(); ++i) {
        if (hist[i] <= hist[i - 1]) continue;
        if (first_peak && noise_buffer < 3) {
            ++noise_buffer;
            continue;
        }
        if (first_peak) {
            first_peak = false;
            max_covg = i;
        } else if (hist[i] > hist[max_covg]) max_covg = i;
    }
    return max_covg;
}

double lognchoosek2(uint32_t n, uint32_t k1, uint32_t k2)
{
    return std::lgamma((double)n + 1) - std::lgamma((double)k1 + 1) - std::lgamma((double)k2 + 1) - std::lgamma((double)n - k1 - k2 + 1);
}

} // namespace

CoverageModel estimate_parameters(const std::vector<uint32_t>& kmer_covg, uint64_t clusters, uint64_t loci_with_clusters, uint32_t global_covg, int k,
    double e_rate, bool bin)
{
    CoverageModel m;
    m.exp_depth_covg = global_covg; // (what pandora returns for a sample without any locus)
    m.e_rate = e_rate;
    m.bin = bin;
    if (loci_with_clusters == 0) {
        m.exp_depth_covg = std::max<uint32_t>(m.exp_depth_covg, 1); // own guard: e = 0 has no likelihood
        m.bin_p = 1.0 / std::exp(m.e_rate * (double)k);
        return m;
    }
    std::vector<uint32_t> hist(COVG_HIST, 0);
    for (uint32_t c : kmer_covg)
        if (c < COVG_HIST) ++hist[c];
    m.num_reads = (uint32_t)(clusters / loci_with_clusters);
    const uint8_t zt = (uint8_t)(global_covg / 10);
    m.mean = fit_mean_covg(hist, zt);
    m.var = fit_variance_covg(hist, m.mean, zt);
    if (m.mean > m.var) { // under-dispersed above covg / 10: look at everything from 2 upwards instead
        m.mean = fit_mean_covg(hist, 2);
        m.var = fit_variance_covg(hist, m.mean, 2);
    }
    if ((bin && m.num_reads > 30 && global_covg > 30) || (!bin && std::fabs(m.var - m.mean) < 2 && m.mean > 10 && m.num_reads > 30 && global_covg > 2)) {
        m.bin = true;
        m.branch = 1;
        const uint32_t peak = find_mean_covg(hist);
        m.exp_depth_covg = peak;
        if (peak > 0 && peak < global_covg) m.e_rate = -std::log((float)peak / (float)global_covg) / (float)k;
    } else if (!bin && m.num_reads > 30 && global_covg > 2 && m.mean < m.var) {
        m.branch = 2;
        // fit_negative_binomial: p = mean / variance, r = (mean p / (1 - p) + variance p^2 / (1 - p)) / 2
        const double p = m.mean / m.var;
        m.nb_p = (float)p;
        m.nb_r = (float)((m.mean * p / (1 - p) + m.var * p * p / (1 - p)) / 2);
        m.exp_depth_covg = (uint32_t)m.mean;
    } else {
        m.branch = 3; // "Insufficient coverage to update error rate"
        m.exp_depth_covg = (uint32_t)fit_mean_covg(hist, zt);
    }
    m.exp_depth_covg = std::max<uint32_t>(m.exp_depth_covg, 1); // (upstream guards branch 3 only; e = 0 has no likelihood in any)
    m.bin_p = 1.0 / std::exp(m.e_rate * (double)k); // set_binomial_parameter_p(e_rate)
    return m;
}

float kmer_log_prob(const CoverageModel& m, uint32_t fwd, uint32_t rev, uint32_t num_reads_of_locus)
{
    const uint32_t c = fwd + rev;
    if (m.bin) { // KmerGraphWithCoverage::bin_prob
        const double p = m.bin_p;
        if (c > num_reads_of_locus) return (float)(lognchoosek2(c, fwd, rev) + (double)c * std::log(p / 2));
        return (float)(lognchoosek2(num_reads_of_locus, fwd, rev) + (double)c * std::log(p / 2) + (double)(num_reads_of_locus - c) * std::log(1 - p));
    }
    // nbin_prob: log pmf of NegativeBinomial(r, p) at c, never below lowest / 1000
    const double r = m.nb_r, p = m.nb_p;
    const double lp = std::lgamma(r + c) - std::lgamma((double)c + 1) - std::lgamma(r) + r * std::log(p) + (double)c * std::log(1 - p);
    return std::max((float)lp, std::numeric_limits<float>::lowest() / 1000);
}

int prob_threshold(const std::vector<float>& log_probs)
{
    // find_prob_thresh: histogram of the integer parts of the log probabilities in (-200, 0]; the threshold is the emptiest
    // bin between its two peaks (the peak of the true k-mers near 0 and the peak of the error k-mers), peaks at least ten
    // bins apart; no second peak: the default (-25) stays
    std::vector<uint32_t> hist((size_t)PROB_HIST, 0);
    for (float p : log_probs)
        if (p >= -(float)PROB_HIST && p < 0) ++hist[(size_t)((int)std::floor(p) + PROB_HIST)];
    int first = 0;
    for (int i = 1; i < PROB_HIST; ++i)
        if (hist[(size_t)i] > hist[(size_t)first]) first = i;
    int second = -1;
    for (int i = 0; i < PROB_HIST; ++i)
        if (std::abs(i - first) >= 10 && hist[(size_t)i] > 0 && (second < 0 || hist[(size_t)i] > hist[(size_t)second])) second = i;
    if (second < 0) return -25;
    const int lo = std::min(first, second), hi = std::max(first, second);
    int at = lo + 1;
    for (int i = lo + 1; i < hi; ++i)
        if (hist[(size_t)i] < hist[(size_t)at]) at = i;
    return at - PROB_HIST;
}

std::vector<uint32_t> find_max_path(const KmerGraph& kg, const std::vector<float>& logp, int thresh, uint32_t max_kmers_to_average)
{
    const uint32_t n = (uint32_t)kg.nodes.size();
    std::vector<uint32_t> path;
    if (n < 3) return path;
    const uint32_t sink = n - 1;
    std::vector<float> best_sum(n, 0.0f);
    std::vector<uint32_t> best_len(n, 0), next(n, sink);
    const float tolerance = 0.000001f;
    for (uint32_t j = n - 1; j != 0; --j) { // node ids are a topological order: n-2 down to the source
        const uint32_t cur = j - 1;
        float max_mean = std::numeric_limits<float>::lowest();
        uint32_t max_length = 0;
        std::vector<uint32_t> succ = kg.nodes[cur].out; // in ascending id order (pandora: whatever order its out-edge vector has)
        std::sort(succ.begin(), succ.end());
        for (uint32_t out : succ) {
            const float avg = best_sum[out] / (float)best_len[out]; // (the sink: 0 / 0, every comparison with it is false)
            const bool terminus = out == sink && (float)thresh > max_mean + tolerance;
            const bool better = avg > max_mean + tolerance;
            const bool close = max_mean - avg <= tolerance;
            const bool longer = best_len[out] > max_length;
            if (!(terminus || better || (close && longer))) continue;
            best_sum[cur] = logp[cur] + best_sum[out];
            best_len[cur] = 1 + best_len[out];
            next[cur] = out;
            if (best_len[cur] > max_kmers_to_average) { // the mean runs over the next max_kmers_to_average nodes only
                uint32_t p = next[cur];
                for (uint32_t step = 0; step < max_kmers_to_average; ++step) p = next[p];
                best_sum[cur] -= logp[p];
                best_len[cur] -= 1;
            }
            if (out != sink) {
                max_mean = best_sum[out] / (float)best_len[out];
                max_length = best_len[out];
            } else max_mean = (float)thresh;
        }
    }
    for (uint32_t p = next[0]; p < sink; p = next[p]) {
        path.push_back(p);
        if (path.size() > 1000000) throw Error(DRPRG_EFORMAT, "find_max_path: the k-mer graph has a cycle");
    }
    return path;
}

std::vector<uint32_t> base_coverage_along_path(const LocalGraph& g, const KmerGraph& kg, const std::vector<uint32_t>& path, const uint32_t* covg)
{
    // local nodes in the order the k-mers reach them (a k-mer path lists its local nodes in walk order; consecutive k-mers of the
    // path overlap), every base starts at coverage 0
    std::vector<uint32_t> order;
    std::vector<int64_t> slot(g.nodes.size(), -1);
    std::vector<std::vector<uint32_t>> per_node;
    for (uint32_t kn : path)
        for (const PathPiece& pc : kg.nodes[kn].path) {
            if (slot[pc.node] < 0) {
                slot[pc.node] = (int64_t)order.size();
                order.push_back(pc.node);
                per_node.emplace_back(g.nodes[pc.node].len(), 0u);
            }
            const uint32_t c = covg[2 * (size_t)kn] + covg[2 * (size_t)kn + 1];
            std::vector<uint32_t>& v = per_node[(size_t)slot[pc.node]];
            for (uint32_t b = pc.off_start; b < pc.off_end && b < v.size(); ++b) v[b] = std::max(v[b], c);
        }
    std::vector<uint32_t> flat;
    for (const auto& v : per_node) flat.insert(flat.end(), v.begin(), v.end());
    return flat;
}

uint32_t mode_u32(std::vector<uint32_t> v)
{
    std::sort(v.begin(), v.end());
    uint32_t counter = 1, max_count = 1, most_common = 0, last = 0;
    bool any = false;
    for (uint32_t x : v) {
        if (any && x == last) ++counter;
        else {
            if (counter > max_count) {
                max_count = counter;
                most_common = last;
            }
            counter = 1;
        }
        last = x;
        any = true;
    }
    if (counter > max_count) most_common = last;
    return most_common;
}

bool path_coverage_too_low(const std::vector<uint32_t>& base_covg, uint32_t global_covg)
{
    if (base_covg.empty()) return false;
    double sum = 0;
    for (uint32_t c : base_covg) sum += c;
    const float mean = (float)(sum / (double)base_covg.size());
    return global_covg > 20 && mode_u32(base_covg) < 3 && mean < 3;
}

} // namespace drprg
