// genotype.cpp -- see genotype.h and DESIGN.md "Semantics".
#include "genotype.h"
#include "params.h"
#include "fastx.h"
#include <algorithm>
#include <cmath>
#include <ctime>
#include <fstream>
#include <functional>
#include <map>

namespace drprg {

std::string format_g(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%g", v);
    return buf;
}

static uint32_t median_u32(std::vector<uint32_t> v)
{
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    size_t n = v.size();
    return (n & 1) ? v[n / 2] : (uint32_t)(((uint64_t)v[n / 2 - 1] + v[n / 2]) / 2);
}

AlleleStats allele_stats(const std::vector<uint32_t>& fwd, const std::vector<uint32_t>& rev, uint32_t min_kmer_covg)
{
    AlleleStats s;
    const size_t n = fwd.size();
    uint64_t sf = 0, sr = 0;
    size_t gaps = 0;
    for (size_t i = 0; i < n; ++i) {
        sf += fwd[i];
        sr += rev[i];
        if (fwd[i] + rev[i] < min_kmer_covg) ++gaps;
    }
    s.sum_fwd = (uint32_t)sf;
    s.sum_rev = (uint32_t)sr;
    s.mean_fwd = n ? (uint32_t)(sf / n) : 0; // integer mean (floor), as in the reference fixtures
    s.mean_rev = n ? (uint32_t)(sr / n) : 0;
    s.med_fwd = median_u32(fwd);
    s.med_rev = median_u32(rev);
    s.gaps = n ? (double)gaps / (double)n : 1.0;
    return s;
}

double allele_likelihood(double e, double c_a, double c_other, double eps, double gaps)
{
    return -e + c_a * std::log(e) - std::lgamma(c_a + 1.0) + c_other * std::log(eps) - e * gaps
        + std::log(1.0 - std::exp(-e)) * (1.0 - gaps);
}

void genotype_site(std::vector<AlleleStats>& alleles, double e, double eps, int& gt, double& gt_conf)
{
    double total = 0;
    for (const AlleleStats& a : alleles) total += (double)a.mean_fwd + (double)a.mean_rev;
    int best = 0;
    for (size_t i = 0; i < alleles.size(); ++i) {
        double c = (double)alleles[i].mean_fwd + (double)alleles[i].mean_rev;
        alleles[i].likelihood = allele_likelihood(e, c, total - c, eps, alleles[i].gaps);
        if (alleles[i].likelihood > alleles[(size_t)best].likelihood) best = (int)i;
    }
    double second = -INFINITY;
    for (size_t i = 0; i < alleles.size(); ++i)
        if ((int)i != best && alleles[i].likelihood > second) second = alleles[i].likelihood;
    gt = best;
    gt_conf = alleles.size() > 1 ? alleles[(size_t)best].likelihood - second : 0.0;
}

uint32_t estimate_exp_depth_covg(const std::vector<uint32_t>& kmer_total_covg, uint32_t zero_thresh)
{
    double sum = 0, cnt = 0;
    for (uint32_t c : kmer_total_covg)
        if (c >= zero_thresh && c < 1000) { sum += c; cnt += 1; }
    uint32_t e = cnt > 0 ? (uint32_t)(sum / cnt) : 0;
    return std::max<uint32_t>(e, 1);
}

std::vector<std::pair<uint32_t, uint32_t>> low_coverage_intervals(const std::vector<uint32_t>& covg, const std::vector<uint8_t>& covered,
    uint32_t min_covg, uint32_t min_len, uint32_t max_len)
{
    std::vector<std::pair<uint32_t, uint32_t>> out;
    const uint32_t n = (uint32_t)covg.size();
    uint32_t i = 0;
    while (i < n) {
        if (!(covered[i] && covg[i] <= min_covg)) {
            ++i;
            continue;
        }
        uint32_t j = i;
        while (j < n && covered[j] && covg[j] <= min_covg) ++j;
        if (j - i >= min_len && j - i <= max_len) out.emplace_back(i, j);
        i = j;
    }
    return out;
}

namespace {

constexpr size_t MAX_ALTS = 10;   // more routes than this through one site -> GRAPHTYPE=TOO_MANY_ALTS
constexpr size_t MAX_ROUTES = 256; // enumeration cap while expanding nested alleles

using Route = std::vector<uint32_t>;

void enumerate_routes(const LocalGraph& g, int chain, std::vector<Route>& out, bool& truncated)
{
    std::vector<Route> res(1);
    const Chain& ch = g.chains[(size_t)chain];
    for (size_t i = 0; i < ch.nodes.size(); ++i) {
        for (Route& r : res) r.push_back(ch.nodes[i]);
        if (i < ch.sites.size()) {
            std::vector<Route> sub;
            for (int al : g.sites[(size_t)ch.sites[i]].alleles) enumerate_routes(g, al, sub, truncated);
            std::vector<Route> next;
            for (const Route& r : res) {
                for (const Route& s : sub) {
                    if (next.size() >= MAX_ROUTES) { truncated = true; break; }
                    Route x = r;
                    x.insert(x.end(), s.begin(), s.end());
                    next.push_back(std::move(x));
                }
            }
            res.swap(next);
        }
    }
    for (Route& r : res) {
        if (out.size() >= MAX_ROUTES) { truncated = true; break; }
        out.push_back(std::move(r));
    }
}

bool chain_has_sites(const LocalGraph& g, int chain) { return !g.chains[(size_t)chain].sites.empty(); }

std::string variant_class(const std::string& ref, const std::string& alt)
{
    if (ref.size() == 1 && alt.size() == 1) return "SNP";
    if (ref.size() == alt.size()) return "PH_SNPs";
    if (ref.size() < alt.size() && alt.compare(0, ref.size(), ref) == 0) return "INDEL";
    if (alt.size() < ref.size() && ref.compare(0, alt.size(), alt) == 0) return "INDEL";
    return "COMPLEX";
}

struct LocusGenotyper {
    const LocalGraph& g;
    const KmerGraph& kg;
    const uint32_t* covg; // this PRG's slice: [2*id], [2*id+1]
    uint32_t knode_base;  // global number of this PRG's k-mer node 0
    const std::vector<uint32_t>& refp;
    const std::string& refseq;
    uint32_t min_kmer_covg;
    double e, eps;
    std::vector<int> ref_index;                    // local node -> index in refp or -1
    std::vector<std::vector<uint32_t>> starts_in;  // local node -> k-mer nodes starting there
    std::vector<VcfRecord>& out;
    struct SiteCall {
        int ref_allele = -1; // chain of the reference allele
        int gt = 0;
        std::vector<Route> alt_routes; // by ALT number - 1
    };
    std::map<int, SiteCall> calls; // every site on the reference path

    void init()
    {
        ref_index.assign(g.nodes.size(), -1);
        for (size_t i = 0; i < refp.size(); ++i) ref_index[refp[i]] = (int)i;
        starts_in.assign(g.nodes.size(), {});
        for (size_t i = 1; i + 1 < kg.nodes.size(); ++i) starts_in[kg.nodes[i].path.front().node].push_back((uint32_t)i);
    }

    // k-mer coverages of the allele whose nodes replace refp(idx_pre, idx_post).  Which k-mers: the k-mer nodes that lie on the route
    // (reference flank + allele + reference flank) and touch the allele AS THE RECORD PRINTS IT -- with its padding base (pad_l: the
    // reference base in front of the site, pad_r: the one behind it; pandora takes the range from the record's POS and the printed
    // allele's length) -- where a k-mer at base s touches [A, B) iff s < B and s + k >= A: the k-mer that ends exactly where the
    // allele starts counts.  Both details are pinned by the reference's fixture VCFs, whose SUM / MEAN / GAPS leak the number of k-mers
    // per allele (tests/golden/kmer_count_kat.tsv, tests/test_kmer_count_kat.py: 316 of 322 informative alleles; with the strict
    // overlap of rounds 1-3, s + k > A on the bare allele, 204 of 262 on in.vcf).
    AlleleStats allele_coverage(int idx_pre, int idx_post, const Route& allele_nodes, std::vector<uint32_t>& knodes, bool pad_l, bool pad_r)
    {
        const int k = kg.k;
        // local route window: > k bases of reference flank on both sides (the padding base and the touching k-mer reach one further)
        Route rt;
        std::vector<int64_t> coord; // sequence coordinate of each route node, allele start = 0
        int lo = idx_pre;
        int64_t have = g.nodes[refp[(size_t)lo]].len();
        while (lo > 0 && have < k + 2) {
            --lo;
            have += g.nodes[refp[(size_t)lo]].len();
        }
        int64_t c = -have;
        for (int i = lo; i <= idx_pre; ++i) {
            rt.push_back(refp[(size_t)i]);
            coord.push_back(c);
            c += g.nodes[refp[(size_t)i]].len();
        }
        // c == 0 here: allele start
        for (uint32_t n : allele_nodes) {
            rt.push_back(n);
            coord.push_back(c);
            c += g.nodes[n].len();
        }
        const int64_t A = pad_l ? -1 : 0, B = c + (pad_r ? 1 : 0);
        int64_t tail = 0;
        for (size_t i = (size_t)idx_post; i < refp.size() && tail < k + 2; ++i) {
            rt.push_back(refp[i]);
            coord.push_back(c);
            c += g.nodes[refp[i]].len();
            tail += g.nodes[refp[i]].len();
        }
        std::vector<uint32_t> fwd, rev;
        for (size_t i = 0; i < rt.size(); ++i) {
            for (uint32_t kn : starts_in[rt[i]]) {
                const KPath& p = kg.nodes[kn].path;
                int64_t s = coord[i] + p.front().off_start;
                if (!(s < B && s + k >= A)) continue; // must touch the printed allele
                if (i + p.size() > rt.size()) continue;
                bool on_route = true;
                for (size_t j = 1; j < p.size(); ++j)
                    if (p[j].node != rt[i + j]) { on_route = false; break; }
                if (!on_route) continue;
                fwd.push_back(covg[2 * (size_t)kn]);
                rev.push_back(covg[2 * (size_t)kn + 1]);
                knodes.push_back(knode_base + kn);
            }
        }
        return allele_stats(fwd, rev, min_kmer_covg);
    }

    void handle_site(int site_id, uint32_t& refpos)
    {
        const Site& st = g.sites[(size_t)site_id];
        int ref_allele = -1;
        for (int al : st.alleles)
            if (ref_index[g.chains[(size_t)al].nodes[0]] >= 0) { ref_allele = al; break; }
        if (ref_allele < 0) throw Error(DRPRG_EFORMAT, "reference path skips a site of " + g.name);
        const uint32_t start = refpos;
        const int idx_pre = ref_index[st.pre_node], idx_post = ref_index[st.post_node];
        walk_chain(ref_allele, refpos); // nested records of the reference allele
        const uint32_t end = refpos;
        Route ref_nodes(refp.begin() + idx_pre + 1, refp.begin() + idx_post);
        std::string ref = refseq.substr(start, end - start);

        bool truncated = false, nested = st.level > 0;
        std::vector<std::pair<std::string, Route>> alts;
        for (int al : st.alleles) {
            if (chain_has_sites(g, al)) nested = true;
            if (al == ref_allele) continue;
            std::vector<Route> routes;
            enumerate_routes(g, al, routes, truncated);
            for (Route& r : routes) {
                std::string s = g.string_along_path(r);
                if (s == ref) continue;
                bool dup = false;
                for (auto& a : alts)
                    if (a.first == s) { dup = true; break; }
                if (!dup) alts.emplace_back(std::move(s), std::move(r));
            }
        }
        calls[site_id].ref_allele = ref_allele;
        if (alts.empty()) return;
        std::sort(alts.begin(), alts.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        if (alts.size() > MAX_ALTS) {
            alts.resize(MAX_ALTS);
            truncated = true;
        }
        // VCF text: pad with the preceding reference base when an allele is empty
        bool any_empty = ref.empty();
        for (auto& a : alts) any_empty |= a.first.empty();
        uint32_t pos0 = start;
        std::string pad_l, pad_r;
        if (any_empty) {
            if (start > 0) {
                pad_l = refseq.substr(start - 1, 1);
                pos0 = start - 1;
            } else if (end < refseq.size()) {
                pad_r = refseq.substr(end, 1);
            }
        }
        VcfRecord rec;
        rec.chrom = g.name;
        rec.allele_knodes.resize(1 + alts.size());
        rec.alleles.push_back(allele_coverage(idx_pre, idx_post, ref_nodes, rec.allele_knodes[0], !pad_l.empty(), !pad_r.empty()));
        for (size_t ai = 0; ai < alts.size(); ++ai)
            rec.alleles.push_back(allele_coverage(idx_pre, idx_post, alts[ai].second, rec.allele_knodes[1 + ai], !pad_l.empty(), !pad_r.empty()));
        genotype_site(rec.alleles, e, eps, rec.gt, rec.gt_conf);
        calls[site_id].gt = rec.gt;
        for (auto& a : alts) calls[site_id].alt_routes.push_back(a.second);
        rec.pos = pos0 + 1;
        rec.ref = pad_l + ref + pad_r;
        for (auto& a : alts) rec.alts.push_back(pad_l + a.first + pad_r);
        rec.vc = variant_class(rec.ref, rec.alts[0]);
        rec.graphtype = truncated ? "TOO_MANY_ALTS" : (nested ? "NESTED" : "SIMPLE");
        out.push_back(std::move(rec));
    }

    // local-node path of the called consensus: the reference path with every site replaced by its called allele
    void consensus_chain(int chain, Route& path) const
    {
        const Chain& ch = g.chains[(size_t)chain];
        for (size_t i = 0; i < ch.nodes.size(); ++i) {
            path.push_back(ch.nodes[i]);
            if (i >= ch.sites.size()) continue;
            auto it = calls.find(ch.sites[i]);
            if (it == calls.end()) throw Error(DRPRG_EFORMAT, "consensus walk left the reference path of " + g.name);
            const SiteCall& c = it->second;
            if (c.gt == 0 || c.alt_routes.empty()) consensus_chain(c.ref_allele, path);
            else path.insert(path.end(), c.alt_routes[(size_t)c.gt - 1].begin(), c.alt_routes[(size_t)c.gt - 1].end());
        }
    }

    // per-base coverage of the consensus (a base takes the largest fwd+rev coverage among the path's k-mer nodes that
    // cover it), low-coverage runs, merged and padded (pandora discover's candidate regions)
    void candidate_regions(const DiscoverParams& dp, uint32_t prg_index, std::vector<CandidateRegion>& regions, std::vector<LocusConsensus>& consensus) const
    {
        Route path;
        consensus_chain(0, path);
        const int k = kg.k;
        std::vector<uint32_t> coord(path.size() + 1, 0);
        for (size_t i = 0; i < path.size(); ++i) coord[i + 1] = coord[i] + g.nodes[path[i]].len();
        const uint32_t L = coord.back();
        std::vector<uint32_t> base(L, 0);
        std::vector<uint8_t> covered(L, 0);
        for (size_t i = 0; i < path.size(); ++i)
            for (uint32_t kn : starts_in[path[i]]) {
                const KPath& p = kg.nodes[kn].path;
                if (i + p.size() > path.size()) continue;
                bool on = true;
                for (size_t j = 1; j < p.size(); ++j)
                    if (p[j].node != path[i + j]) { on = false; break; }
                if (!on) continue;
                const uint32_t s = coord[i] + p.front().off_start, c = covg[2 * (size_t)kn] + covg[2 * (size_t)kn + 1];
                for (uint32_t b = s; b < s + (uint32_t)k && b < L; ++b) {
                    base[b] = covered[b] ? std::max(base[b], c) : c;
                    covered[b] = 1;
                }
            }
        auto runs = low_coverage_intervals(base, covered, dp.min_candidate_covg, dp.min_candidate_len, dp.max_candidate_len);
        if (runs.empty()) return;
        const std::string cons = g.string_along_path(path);
        // merge runs closer than merge_dist, then pad
        std::vector<std::pair<uint32_t, uint32_t>> merged;
        for (auto& r : runs) {
            if (!merged.empty() && r.first <= merged.back().second + dp.merge_dist) merged.back().second = r.second;
            else merged.push_back(r);
        }
        {
            LocusConsensus lc;
            lc.chrom = g.name;
            lc.prg = prg_index;
            lc.seq = cons;
            for (uint32_t n : path) lc.nodes.push_back(ConsensusNode { n, g.nodes[n].start, g.nodes[n].end, g.nodes[n].seq });
            consensus.push_back(std::move(lc));
        }
        for (auto& r : merged) {
            CandidateRegion cr;
            cr.chrom = g.name;
            cr.prg = prg_index;
            cr.low_start = r.first;
            cr.low_end = r.second;
            cr.start = r.first > dp.padding ? r.first - dp.padding : 0;
            cr.end = std::min<uint32_t>(L, r.second + dp.padding);
            for (uint32_t b = r.first; b < r.second; ++b)
                if (covered[b]) cr.max_covg = std::max(cr.max_covg, base[b]);
            cr.seq = cons.substr(cr.start, cr.end - cr.start);
            if (cr.start >= dp.anchor_len) cr.left_anchor = cons.substr(cr.start - dp.anchor_len, dp.anchor_len);
            if (cr.end + dp.anchor_len <= L) cr.right_anchor = cons.substr(cr.end, dp.anchor_len);
            if (dp.anchor_len) {
                const uint32_t nl = std::min<uint32_t>(3, cr.start / dp.anchor_len), nr = std::min<uint32_t>(3, (uint32_t)(L - cr.end) / dp.anchor_len);
                cr.left_context = cons.substr(cr.start - nl * dp.anchor_len, nl * dp.anchor_len);
                cr.right_context = cons.substr(cr.end, nr * dp.anchor_len);
            }
            regions.push_back(std::move(cr));
        }
    }

    void walk_chain(int chain, uint32_t& refpos)
    {
        const Chain& ch = g.chains[(size_t)chain];
        for (size_t i = 0; i < ch.nodes.size(); ++i) {
            refpos += g.nodes[ch.nodes[i]].len();
            if (i < ch.sites.size()) handle_site(ch.sites[i], refpos);
        }
    }
};

} // namespace

GenotypeResult genotype(const PrgIndex& idx, const std::vector<uint32_t>& covg, const std::vector<uint32_t>& prg_reads,
    uint64_t total_bases, const MapParams& p, const std::string& vcf_refs, const DiscoverParams& dp)
{
    const FlatIndex& f = idx.flat;
    if (covg.size() != 2 * (size_t)f.total_knodes() || prg_reads.size() != idx.prgs.size())
        throw Error(DRPRG_EINVAL, "coverage vector does not match the index");
    GenotypeResult res;
    std::map<std::string, std::string> refs;
    if (!vcf_refs.empty())
        for (auto& kv : read_fasta(vcf_refs)) refs[kv.first] = kv.second;

    // ---- the coverage model of the sample over the loci that have clusters (pandora estimate_parameters; params.cpp) ----
    std::vector<bool> present(idx.prgs.size(), false);
    std::vector<uint32_t> kcov;
    uint64_t clusters = 0, loci = 0;
    auto sat = [](uint32_t c) { return std::min<uint32_t>(c, 65535u); }; // pandora keeps u16 saturating counters
    for (size_t pi = 0; pi < idx.prgs.size(); ++pi) {
        if (prg_reads[pi] == 0) continue;
        clusters += prg_reads[pi];
        ++loci;
        const uint32_t base = f.knode_base[pi], n = (uint32_t)idx.kgs[pi].nodes.size();
        for (uint32_t i = 1; i + 1 < n; ++i) kcov.push_back(sat(covg[2 * (size_t)(base + i)]) + sat(covg[2 * (size_t)(base + i) + 1]));
    }
    const uint32_t global_covg = (uint32_t)std::min<uint64_t>(total_bases / std::max<uint64_t>(p.genome_size, 1), 0xFFFFFFFFull);
    CoverageModel model = estimate_parameters(kcov, clusters, loci, global_covg, p.k, p.error_rate, p.binomial);
    {
        std::vector<float> lp;
        lp.reserve(kcov.size());
        for (size_t pi = 0; pi < idx.prgs.size(); ++pi) {
            if (prg_reads[pi] == 0) continue;
            const uint32_t base = f.knode_base[pi], n = (uint32_t)idx.kgs[pi].nodes.size();
            for (uint32_t i = 1; i + 1 < n; ++i)
                lp.push_back(kmer_log_prob(model, sat(covg[2 * (size_t)(base + i)]), sat(covg[2 * (size_t)(base + i) + 1]), prg_reads[pi]));
        }
        model.thresh = prob_threshold(lp);
    }
    res.exp_depth_covg = model.exp_depth_covg;
    res.min_kmer_covg = res.exp_depth_covg / 10;
    res.model = model;
    // ---- which loci stay: one with clusters whose maximum-likelihood path is not almost bare in a deep sample
    // (LocalPRG::add_consensus_path_to_fastaq clears the path, pandora map then removes the node: no ##contig line) ----
    for (size_t pi = 0; pi < idx.prgs.size(); ++pi) {
        if (prg_reads[pi] == 0) continue;
        const KmerGraph& kg = idx.kgs[pi];
        const uint32_t base = f.knode_base[pi], n = (uint32_t)kg.nodes.size();
        std::vector<uint32_t> local(2 * (size_t)n);
        for (size_t i = 0; i < local.size(); ++i) local[i] = sat(covg[2 * (size_t)base + i]);
        std::vector<float> lp(n, 0.0f);
        for (uint32_t i = 1; i + 1 < n; ++i) lp[i] = kmer_log_prob(model, local[2 * (size_t)i], local[2 * (size_t)i + 1], prg_reads[pi]);
        const std::vector<uint32_t> mlp = find_max_path(kg, lp, model.thresh);
        if (mlp.empty()) continue; // no path: nothing to call on
        if (path_coverage_too_low(base_coverage_along_path(idx.prgs[pi], kg, mlp, local.data()), global_covg)) {
            res.dropped_low_coverage.push_back(idx.prgs[pi].name);
            continue;
        }
        present[pi] = true;
    }

    for (size_t pi = 0; pi < idx.prgs.size(); ++pi) {
        const LocalGraph& g = idx.prgs[pi];
        if (!present[pi]) {
            res.absent.push_back(g.name);
            continue;
        }
        res.present.push_back(g.name);
        std::vector<uint32_t> refp;
        std::string refseq;
        auto it = refs.find(g.name);
        if (it != refs.end()) {
            refp = g.nodes_along_string(it->second);
            if (refp.empty())
                std::fprintf(stderr, "[drprg-hip] warning: --vcf-refs sequence of %s is not a path of its PRG; using the first-allele path\n",
                    g.name.c_str());
        }
        if (refp.empty()) refp = g.top_path();
        refseq = g.string_along_path(refp);
        // saturate like pandora's u16 coverage counters
        const uint32_t base = f.knode_base[pi], n = (uint32_t)idx.kgs[pi].nodes.size();
        std::vector<uint32_t> local(2 * (size_t)n);
        for (size_t i = 0; i < local.size(); ++i) local[i] = std::min<uint32_t>(covg[2 * (size_t)base + i], 65535u);
        LocusGenotyper lg { g, idx.kgs[pi], local.data(), base, refp, refseq, res.min_kmer_covg, (double)res.exp_depth_covg,
            p.genotyping_error_rate, {}, {}, res.records, {} };
        lg.init();
        uint32_t refpos = 0;
        lg.walk_chain(0, refpos);
        lg.candidate_regions(dp, (uint32_t)pi, res.candidates, res.consensus);
    }
    std::sort(res.present.begin(), res.present.end());
    std::sort(res.absent.begin(), res.absent.end());
    std::sort(res.records.begin(), res.records.end(), [](const VcfRecord& a, const VcfRecord& b) {
        if (a.chrom != b.chrom) return a.chrom < b.chrom;
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.ref != b.ref) return a.ref < b.ref;
        return a.alts < b.alts;
    });
    return res;
}

void write_vcf(const std::string& path, const GenotypeResult& r, const std::string& sample)
{
    std::ofstream o(path);
    if (!o) throw Error(DRPRG_EIO, "cannot write " + path);
    char date[32];
    std::time_t t = std::time(nullptr);
    std::strftime(date, sizeof date, "%d/%m/%y", std::localtime(&t));
    o << "##fileformat=VCFv4.3\n";
    o << "##fileDate==" << date << "\n";
    o << "##ALT=<ID=SNP,Description=\"SNP\">\n";
    o << "##ALT=<ID=PH_SNPs,Description=\"Phased SNPs\">\n";
    o << "##ALT=<ID=INDEL,Description=\"Insertion-deletion\">\n";
    o << "##ALT=<ID=COMPLEX,Description=\"Complex variant, collection of SNPs and indels\">\n";
    o << "##INFO=<ID=VC,Number=1,Type=String,Description=\"Type (class) of variant\">\n";
    o << "##ALT=<ID=SIMPLE,Description=\"Graph bubble is simple\">\n";
    o << "##ALT=<ID=NESTED,Description=\"Variation site was a nested feature in the graph\">\n";
    o << "##ALT=<ID=TOO_MANY_ALTS,Description=\"Variation site was a multinested feature with too many alts to include all in the VCF\">\n";
    o << "##INFO=<ID=GRAPHTYPE,Number=1,Type=String,Description=\"Type of graph feature\">\n";
    o << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
    o << "##FORMAT=<ID=MEAN_FWD_COVG,Number=R,Type=Integer,Description=\"Mean forward coverage\">\n";
    o << "##FORMAT=<ID=MEAN_REV_COVG,Number=R,Type=Integer,Description=\"Mean reverse coverage\">\n";
    o << "##FORMAT=<ID=MED_FWD_COVG,Number=R,Type=Integer,Description=\"Med forward coverage\">\n";
    o << "##FORMAT=<ID=MED_REV_COVG,Number=R,Type=Integer,Description=\"Med reverse coverage\">\n";
    o << "##FORMAT=<ID=SUM_FWD_COVG,Number=R,Type=Integer,Description=\"Sum forward coverage\">\n";
    o << "##FORMAT=<ID=SUM_REV_COVG,Number=R,Type=Integer,Description=\"Sum reverse coverage\">\n";
    o << "##FORMAT=<ID=GAPS,Number=R,Type=Float,Description=\"Number of gap bases\">\n";
    o << "##FORMAT=<ID=LIKELIHOOD,Number=R,Type=Float,Description=\"Likelihood\">\n";
    o << "##FORMAT=<ID=GT_CONF,Number=1,Type=Float,Description=\"Genotype confidence\">\n";
    for (const std::string& c : r.present) o << "##contig=<ID=" << c << ">\n";
    o << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" << sample << "\n";
    auto join = [&](const VcfRecord& rec, auto get) {
        std::string s;
        for (size_t i = 0; i < rec.alleles.size(); ++i) {
            if (i) s += ",";
            s += get(rec.alleles[i]);
        }
        return s;
    };
    for (const VcfRecord& rec : r.records) {
        o << rec.chrom << "\t" << rec.pos << "\t.\t" << rec.ref << "\t";
        for (size_t i = 0; i < rec.alts.size(); ++i) o << (i ? "," : "") << rec.alts[i];
        o << "\t.\t.\tVC=" << rec.vc << ";GRAPHTYPE=" << rec.graphtype
          << "\tGT:MEAN_FWD_COVG:MEAN_REV_COVG:MED_FWD_COVG:MED_REV_COVG:SUM_FWD_COVG:SUM_REV_COVG:GAPS:LIKELIHOOD:GT_CONF\t";
        o << rec.gt;
        o << ":" << join(rec, [](const AlleleStats& a) { return std::to_string(a.mean_fwd); });
        o << ":" << join(rec, [](const AlleleStats& a) { return std::to_string(a.mean_rev); });
        o << ":" << join(rec, [](const AlleleStats& a) { return std::to_string(a.med_fwd); });
        o << ":" << join(rec, [](const AlleleStats& a) { return std::to_string(a.med_rev); });
        o << ":" << join(rec, [](const AlleleStats& a) { return std::to_string(a.sum_fwd); });
        o << ":" << join(rec, [](const AlleleStats& a) { return std::to_string(a.sum_rev); });
        o << ":" << join(rec, [](const AlleleStats& a) { return format_g(a.gaps); });
        o << ":" << join(rec, [](const AlleleStats& a) { return format_g(a.likelihood); });
        o << ":" << format_g(rec.gt_conf) << "\n";
    }
    if (!o) throw Error(DRPRG_EIO, "short write to " + path);
}

} // namespace drprg
