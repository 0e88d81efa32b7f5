/*
 * oracle_params.c -- TEST INFRASTRUCTURE (the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load anything under oracle/).
 *
 * A second, separately written statement of what `pandora map` computes between its read loop and the VCF
 * (reference call site /root/reference/src/lib.rs:580-642; consumers: e in every LIKELIHOOD / GT_CONF,
 * /root/reference/src/filter.rs:12-16 and :149; the ##contig lines, /root/reference/src/predict.rs:757-765):
 *   - estimate_parameters (pandora src/estimate_parameters.cpp): k-mer coverage histogram -> model + exp_depth_covg,
 *   - the k-mer log probabilities (src/kmergraphwithcoverage.cpp nbin_prob / bin_prob) and the threshold find_prob_thresh derives,
 *   - find_max_path (same file) and the rule that drops a locus whose best path is almost bare (src/localPRG.cpp
 *     add_consensus_path_to_fastaq; mode() and mean() of src/utils.cpp).
 * PARITY UNPINNED: pandora's source is not under /root/reference and no reference fixture exercises these functions; the only
 * thing the seven fixture VCFs say is that e is a positive integer (tests/test_params.py records which).  Written from knowledge of
 * upstream pandora 0.9 / 0.10 [UPSTREAM-MEMORY], in plain loops and flat arrays, sharing nothing with drprg_amd/csrc/params.cpp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* moments of hist[zero_thresh ..] (fit_mean_covg, fit_variance_covg); the threshold travels in 8 bits upstream */
static double moment_mean(const uint32_t* hist, int n, unsigned zero_thresh)
{
    double num = 0, den = 0;
    for (int i = (int)(zero_thresh & 0xFFu); i < n; ++i) {
        num += (double)i * hist[i];
        den += hist[i];
    }
    return den > 0 ? num / den : 0;
}
static double moment_var(const uint32_t* hist, int n, double mean, unsigned zero_thresh)
{
    double num = 0, den = 0;
    for (int i = (int)(zero_thresh & 0xFFu); i < n; ++i) {
        num += ((double)i - mean) * ((double)i - mean) * hist[i];
        den += hist[i];
    }
    return den > 0 ? num / den : 0;
}

/* find_mean_covg: top of the histogram past the error peak; three rises are written off as noise, the fourth ends the first peak */
static uint32_t second_peak(const uint32_t* hist, int n)
{
    int in_first = 1, rises = 0;
    uint32_t best = 0;
    for (int i = 1; i < n; ++i) {
        if (hist[i] <= hist[i - 1]) continue;
        if (in_first) {
            if (rises < 3) ++rises;
            else {
                in_first = 0;
                best = (uint32_t)i;
            }
        } else if (hist[i] > hist[best]) best = (uint32_t)i;
    }
    return best;
}

/*
 * out[0] exp_depth_covg, [1] binomial model in force (0/1), [2] e_rate, [3] nb_p, [4] nb_r, [5] branch (1 binomial, 2 negative
 * binomial, 3 insufficient coverage, 0 no locus), [6] mean, [7] variance, [8] reads (clusters) per locus, [9] binomial p
 */
ORC_API void orc_estimate_parameters(const uint32_t* kmer_covg, int64_t n, uint64_t clusters, uint64_t loci, uint32_t global_covg, int k,
    double e_rate, int bin, double* out)
{
    double e = e_rate, nb_p = 0.015f, nb_r = 2.0f, mean = 0, var = 0;
    uint32_t exp_depth = global_covg, reads = 0;
    int branch = 0, use_bin = bin != 0;
    if (loci > 0) {
        uint32_t hist[1000];
        memset(hist, 0, sizeof hist);
        for (int64_t i = 0; i < n; ++i)
            if (kmer_covg[i] < 1000) hist[kmer_covg[i]]++;
        reads = (uint32_t)(clusters / loci);
        mean = moment_mean(hist, 1000, global_covg / 10);
        var = moment_var(hist, 1000, mean, global_covg / 10);
        if (mean > var) {
            mean = moment_mean(hist, 1000, 2);
            var = moment_var(hist, 1000, mean, 2);
        }
        const int deep = reads > 30;
        if ((bin && deep && global_covg > 30) || (!bin && fabs(var - mean) < 2 && mean > 10 && deep && global_covg > 2)) {
            branch = 1;
            use_bin = 1;
            exp_depth = second_peak(hist, 1000);
            if (exp_depth > 0 && exp_depth < global_covg) e = -logf((float)exp_depth / (float)global_covg) / (float)k;
        } else if (!bin && deep && global_covg > 2 && mean < var) {
            branch = 2;
            const double p = mean / var;
            nb_p = (float)p;
            nb_r = (float)((mean * p / (1 - p) + var * p * p / (1 - p)) / 2);
            exp_depth = (uint32_t)mean;
        } else {
            branch = 3;
            exp_depth = (uint32_t)moment_mean(hist, 1000, global_covg / 10);
        }
    }
    if (exp_depth < 1) exp_depth = 1;
    out[0] = exp_depth;
    out[1] = use_bin;
    out[2] = e;
    out[3] = nb_p;
    out[4] = nb_r;
    out[5] = branch;
    out[6] = mean;
    out[7] = var;
    out[8] = reads;
    out[9] = 1.0 / exp(e * (double)k);
}

static double ln_choose2(double n, double a, double b) { return lgamma(n + 1) - lgamma(a + 1) - lgamma(b + 1) - lgamma(n - a - b + 1); }

/* log probability of one k-mer's coverage: negative binomial (use_bin = 0: nb_p, nb_r) or binomial (bin_p, the locus' reads) */
ORC_API float orc_kmer_log_prob(int use_bin, double nb_p, double nb_r, double bin_p, uint32_t fwd, uint32_t rev, uint32_t locus_reads)
{
    const double c = (double)fwd + (double)rev;
    if (use_bin) {
        if (c > (double)locus_reads) return (float)(ln_choose2(c, fwd, rev) + c * log(bin_p / 2));
        return (float)(ln_choose2(locus_reads, fwd, rev) + c * log(bin_p / 2) + ((double)locus_reads - c) * log(1 - bin_p));
    }
    const float v = (float)(lgamma(nb_r + c) - lgamma(c + 1) - lgamma(nb_r) + nb_r * log(nb_p) + c * log(1 - nb_p));
    const float floor_v = -FLT_MAX / 1000;
    return v > floor_v ? v : floor_v;
}

/* the emptiest bin between the two peaks (>= 10 bins apart) of the histogram of floor(log p) over [-200, 0); -25 without a second peak */
ORC_API int orc_prob_threshold(const float* logp, int64_t n)
{
    uint32_t bins[200];
    memset(bins, 0, sizeof bins);
    for (int64_t i = 0; i < n; ++i)
        if (logp[i] >= -200.0f && logp[i] < 0.0f) bins[(int)floorf(logp[i]) + 200]++;
    int a = 0, b = -1;
    for (int i = 0; i < 200; ++i)
        if (bins[i] > bins[a]) a = i;
    for (int i = 0; i < 200; ++i) {
        if (abs(i - a) < 10 || bins[i] == 0) continue;
        if (b < 0 || bins[i] > bins[b]) b = i;
    }
    if (b < 0) return -25;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    int m = lo + 1;
    for (int i = lo + 2; i < hi; ++i)
        if (bins[i] < bins[m]) m = i;
    return m - 200;
}

typedef struct {
    uint32_t from, to;
} edge_t;
static int edge_cmp(const void* x, const void* y)
{
    const edge_t *a = (const edge_t*)x, *b = (const edge_t*)y;
    if (a->from != b->from) return a->from < b->from ? -1 : 1;
    return a->to < b->to ? -1 : (a->to > b->to ? 1 : 0);
}

/*
 * find_max_path over a k-mer graph given as an edge list in node-id space (0 = source, n_nodes - 1 = sink, ids topologically
 * ordered).  Successors are looked at in ascending id order (pandora: the order its out-edge vector happens to have).  Returns the
 * number of nodes on the path (source and sink excluded), the first `cap` of them in path[].
 */
ORC_API int64_t orc_max_path(uint32_t n_nodes, uint64_t n_edges, const uint32_t* from, const uint32_t* to, const float* logp, int thresh,
    uint32_t max_avg, uint32_t* path, int64_t cap)
{
    if (n_nodes < 3) return 0;
    edge_t* e = (edge_t*)malloc(sizeof(edge_t) * (size_t)(n_edges ? n_edges : 1));
    for (uint64_t i = 0; i < n_edges; ++i) {
        e[i].from = from[i];
        e[i].to = to[i];
    }
    qsort(e, (size_t)n_edges, sizeof(edge_t), edge_cmp);
    uint64_t* first = (uint64_t*)calloc((size_t)n_nodes + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < n_edges; ++i) first[e[i].from + 1]++;
    for (uint32_t v = 0; v < n_nodes; ++v) first[v + 1] += first[v];
    float* sum = (float*)calloc(n_nodes, sizeof(float));
    uint32_t* len = (uint32_t*)calloc(n_nodes, sizeof(uint32_t));
    uint32_t* nxt = (uint32_t*)malloc(sizeof(uint32_t) * n_nodes);
    const uint32_t sink = n_nodes - 1;
    for (uint32_t v = 0; v < n_nodes; ++v) nxt[v] = sink;
    const float tol = 0.000001f;
    for (uint32_t v = n_nodes - 1; v-- > 0;) {
        float top = -FLT_MAX;
        uint32_t top_len = 0;
        for (uint64_t i = first[v]; i < first[v + 1]; ++i) {
            const uint32_t o = e[i].to;
            const float avg = sum[o] / (float)len[o];
            int take = 0;
            if (o == sink && (float)thresh > top + tol) take = 1;
            else if (avg > top + tol) take = 1;
            else if (top - avg <= tol && len[o] > top_len) take = 1;
            if (!take) continue;
            sum[v] = logp[v] + sum[o];
            len[v] = len[o] + 1;
            nxt[v] = o;
            if (len[v] > max_avg) {
                uint32_t q = nxt[v];
                for (uint32_t s = 0; s < max_avg; ++s) q = nxt[q];
                sum[v] -= logp[q];
                len[v]--;
            }
            if (o == sink) top = (float)thresh;
            else {
                top = sum[o] / (float)len[o];
                top_len = len[o];
            }
        }
    }
    int64_t n = 0;
    for (uint32_t q = nxt[0]; q < sink && n < 2000000; q = nxt[q]) {
        if (n < cap) path[n] = q;
        ++n;
    }
    free(e);
    free(first);
    free(sum);
    free(len);
    free(nxt);
    return n;
}

static int u32_cmp(const void* a, const void* b)
{
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* pandora utils.cpp mode(): the smallest of the most frequent values if it occurs at least twice, else 0 */
ORC_API uint32_t orc_mode(const uint32_t* v, int64_t n)
{
    if (n <= 0) return 0;
    uint32_t* t = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    memcpy(t, v, sizeof(uint32_t) * (size_t)n);
    qsort(t, (size_t)n, sizeof(uint32_t), u32_cmp);
    uint32_t best = 0, best_run = 1;
    int64_t i = 0;
    while (i < n) {
        int64_t j = i;
        while (j < n && t[j] == t[i]) ++j;
        if ((uint32_t)(j - i) > best_run) {
            best_run = (uint32_t)(j - i);
            best = t[i];
        }
        i = j;
    }
    free(t);
    return best;
}

/* 1: the locus is dropped (deep sample, mode and mean of the per-base coverage of its best path both below 3) */
ORC_API int orc_path_coverage_too_low(const uint32_t* base_covg, int64_t n, uint32_t global_covg)
{
    if (n <= 0) return 0;
    double s = 0;
    for (int64_t i = 0; i < n; ++i) s += base_covg[i];
    return global_covg > 20 && orc_mode(base_covg, n) < 3 && (float)(s / (double)n) < 3.0f;
}
