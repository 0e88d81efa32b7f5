// denovo.cpp -- the second half of `pandora discover` (SURVEY.md section 8f NEXT-2; reference call site
// /root/reference/src/lib.rs:513-578, consumer /root/reference/src/predict.rs:247-284): what do the reads say inside the
// candidate regions (genotype.cpp candidate_regions) that the called consensus does not?
//
// pandora assembles the reads of a region with a de Bruijn graph (GATB); its source is not in the reference tree.  This
// is a simpler, exact-anchor pile-up that serves accurate (Illumina) reads: the region's padding guarantees well-covered
// consensus on both sides, so the anchor_len bases before and after the padded region are looked up, as exact k-mers in
// either orientation, in every read of the file (a second, host-side pass over the reads: the mapping pass keeps no
// reads); a read that holds both anchors of a region in the right order spells one allele between them; the most
// frequent allele that is not the consensus, with at least min_support reads and min_fraction of the spanning reads, is
// a novel variant.  Noisy long reads (no -I) spell the allele with their own errors: their strings are aligned to the consensus
// slice and the majority is taken column by column instead (column_consensus).
//
// Round 4: LOCAL ASSEMBLY beside the pile-up, for accurate reads (assemble_region below): every read that shares a k-mer with a region's
// slice of the consensus (anchors included) joins the region's pile, oriented like the consensus; the pile's k-mers (k = anchor_len = 15,
// pandora's --discover-k default [UPSTREAM-MEMORY]) seen at least min_dbg_dp = 2 times [UPSTREAM-MEMORY: --min-dbg-dp] are the nodes of a
// de Bruijn graph; paths are searched depth first from a k-mer of the left flank to a k-mer of the right flank (outermost first, as
// pandora's start / end k-mer lists), no longer than the consensus between them + max_len_change, at most 25 of them; every path that
// is not the consensus is an allele.  It finds what the pile-up cannot see -- reads too short to hold both anchors, minor alleles of a
// mixed sample -- and it is the method family of the reference; the pile-up stays the source of the majority allele's read counts.
// DRPRG_HIP_DENOVO=pileup | dbg | both (default both: the pile-up's variant, then every assembled allele that is not that variant).
//
// Outputs: denovo_paths.txt in the layout the reference's parser and make_prg read (/root/reference/src/lib.rs:648-697 and
// the example at :3010-3038: locus, "<n> nodes", one "(id [start, end) seq)" line per local node of the called path,
// "<m> denovo variants for this locus", "pos<TAB>ref<TAB>alt" lines -- positions 1-based on the path sequence
// [UPSTREAM-MEMORY]); and, for hosts without make_prg / mafft, the PRG updated in place: a variant that lies inside ONE local
// node becomes a new site of that PRG (MakePrg::update's job in the reference, /root/reference/src/lib.rs:279-456).
#include "denovo.h"
#include "fastx.h"
#include "ingest.h"
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <unordered_map>

namespace drprg {

namespace {

struct AnchorRef {
    uint32_t region;
    uint8_t right;   // 0 = left anchor, 1 = right anchor
    uint8_t reverse; // the reverse complement of the anchor (the read runs against the consensus)
    uint8_t j;       // 0 = the anchor next to the region, 1, 2 = the fallback anchors further out (noisy reads only)
};

bool pack_kmer(const char* s, uint32_t k, uint64_t& out)
{
    uint64_t v = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const int c = nt4((unsigned char)s[i]);
        if (c > 3) return false;
        v = (v << 2) | (uint64_t)c;
    }
    out = v;
    return true;
}

uint32_t edit_distance(const std::string& a, const std::string& b)
{
    std::vector<uint32_t> prev(b.size() + 1), cur(b.size() + 1);
    for (size_t j = 0; j <= b.size(); ++j) prev[j] = (uint32_t)j;
    for (size_t i = 1; i <= a.size(); ++i) {
        cur[0] = (uint32_t)i;
        for (size_t j = 1; j <= b.size(); ++j) cur[j] = std::min(prev[j - 1] + (a[i - 1] != b[j - 1]), std::min(prev[j], cur[j - 1]) + 1);
        prev.swap(cur);
    }
    return prev[b.size()];
}

std::string revcomp(const std::string& s)
{
    std::string r(s.rbegin(), s.rend());
    for (char& c : r) {
        switch (c) {
        case 'A': c = 'T'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        case 'T': c = 'A'; break;
        default: c = 'N';
        }
    }
    return r;
}

struct RegionVotes {
    std::mutex mu;
    std::unordered_map<std::string, uint32_t> alleles;
};

// votes are keyed by two digits (which left / right anchor the read held: 0 = the one next to the region) + the string spelled
// between the two anchors; the consensus that string is to be compared with is the region plus the skipped anchors
std::string extended(const CandidateRegion& c, uint32_t A, uint32_t jl, uint32_t jr, const std::string& core)
{
    return c.left_context.substr(c.left_context.size() - (size_t)A * jl) + core + c.right_context.substr(0, (size_t)A * jr);
}

// Noisy reads (no -I): the strings the reads spell between the anchors differ from each other by their own errors, so they
// are not counted as whole strings but column by column: every string is aligned to the consensus slice (global, unit costs),
// a column of the region collects votes for A / C / G / T / "deleted", the gap before a column collects the inserted strings,
// and the majority of every column -- if it has min_support reads and min_fraction of the spanning reads -- makes the allele.
std::string column_consensus(const CandidateRegion& cr, const std::string& core, uint32_t A, const std::unordered_map<std::string, uint32_t>& alleles,
    uint32_t spanning, const DiscoverParams& dp)
{
    const size_t R = core.size();
    std::vector<std::array<uint32_t, 5>> col(R, std::array<uint32_t, 5> { 0, 0, 0, 0, 0 }); // A C G T deleted
    std::vector<std::unordered_map<std::string, uint32_t>> ins(R + 1);                        // inserted before column i
    std::vector<uint16_t> dpm;
    for (auto& kv : alleles) {
        const uint32_t jl = (uint32_t)(kv.first[0] - '0'), jr = (uint32_t)(kv.first[1] - '0');
        const std::string ref = extended(cr, A, jl, jr, core), s = kv.first.substr(2);
        const size_t E = ref.size(), S = s.size(), shift = (size_t)A * jl; // region column = extended column - shift
        dpm.assign((E + 1) * (S + 1), 0);
        auto D = [&](size_t i, size_t j) -> uint16_t& { return dpm[i * (S + 1) + j]; };
        for (size_t i = 0; i <= E; ++i) D(i, 0) = (uint16_t)i;
        for (size_t j = 0; j <= S; ++j) D(0, j) = (uint16_t)j;
        for (size_t i = 1; i <= E; ++i)
            for (size_t j = 1; j <= S; ++j) {
                const uint16_t sub = (uint16_t)(D(i - 1, j - 1) + (ref[i - 1] != s[j - 1])), del = (uint16_t)(D(i - 1, j) + 1), in = (uint16_t)(D(i, j - 1) + 1);
                D(i, j) = std::min(sub, std::min(del, in));
            }
        // traceback from the end; on ties prefer the diagonal, then a deletion: gaps end up as far left as the costs allow
        size_t i = E, j = S;
        std::string pending; // bases inserted before extended column i (collected right to left)
        auto flush_ins = [&](size_t before_col) {
            if (pending.empty()) return;
            if (before_col >= shift && before_col - shift <= R) {
                std::reverse(pending.begin(), pending.end());
                ins[before_col - shift][pending] += kv.second;
            }
            pending.clear();
        };
        auto in_region = [&](size_t ext_col) { return ext_col >= shift && ext_col - shift < R; };
        while (i > 0 || j > 0) {
            if (i > 0 && j > 0 && D(i, j) == D(i - 1, j - 1) + (ref[i - 1] != s[j - 1])) {
                flush_ins(i);
                const int c = nt4((unsigned char)s[j - 1]);
                if (c < 4 && in_region(i - 1)) col[i - 1 - shift][(size_t)c] += kv.second;
                --i;
                --j;
            } else if (i > 0 && D(i, j) == D(i - 1, j) + 1) {
                flush_ins(i);
                if (in_region(i - 1)) col[i - 1 - shift][4] += kv.second;
                --i;
            } else {
                pending.push_back(s[j - 1]);
                --j;
            }
        }
        flush_ins(0);
    }
    const uint32_t need = std::max<uint32_t>(dp.min_support, (uint32_t)std::ceil(dp.min_fraction * (double)spanning));
    std::string out;
    for (size_t i = 0; i <= R; ++i) {
        // an insertion: enough reads insert SOMETHING here; its length is the most frequent one, its bases the majority per place
        uint32_t any = 0;
        std::map<size_t, uint32_t> by_len;
        for (auto& kv : ins[i]) {
            any += kv.second;
            by_len[kv.first.size()] += kv.second;
        }
        if (any >= need) {
            size_t len = 0;
            uint32_t len_n = 0;
            for (auto& kv : by_len)
                if (kv.second > len_n) {
                    len_n = kv.second;
                    len = kv.first;
                }
            for (size_t p = 0; p < len; ++p) {
                uint32_t n[4] = { 0, 0, 0, 0 };
                for (auto& kv : ins[i])
                    if (kv.first.size() == len && nt4((unsigned char)kv.first[p]) < 4) n[nt4((unsigned char)kv.first[p])] += kv.second;
                out.push_back("ACGT"[std::max_element(n, n + 4) - n]);
            }
        }
        if (i == R) break;
        const size_t ref_c = (size_t)nt4((unsigned char)core[i]);
        size_t arg = ref_c > 3 ? 0 : ref_c;
        for (size_t c = 0; c < 5; ++c)
            if (col[i][c] > col[i][arg]) arg = c;
        if (arg != ref_c && col[i][arg] < need) arg = ref_c; // a change needs the support; otherwise the consensus base stays
        if (arg < 4) out.push_back("ACGT"[arg]);
    }
    return out;
}

// ---- local assembly of one region (accurate reads) ---------------------------------------------------------------------------------
struct AssembledAllele {
    std::string slice;   // the region's slice of the consensus (left anchor + region + right anchor) as this path spells it
    uint32_t support;    // smallest count among the path's k-mers that the consensus slice does not hold
};
constexpr uint32_t DBG_MIN_DP = 2, DBG_MAX_PATHS = 25, DBG_MAX_STEPS = 200000;

// slice: left anchor + region + right anchor; flank_l / flank_r: how many bases on either side are anchor + padding (start / end k-mers
// lie inside them); pile: the reads that share a k-mer with the slice, oriented like it.  Returns the alleles, best supported first; `depth`:
// the largest count of a consensus k-mer of the flanks (what "spanning" is for an assembled allele).
std::vector<AssembledAllele> assemble_region(const std::string& slice, uint32_t flank_l, uint32_t flank_r, const std::vector<std::string>& pile, uint32_t K,
    uint32_t max_len_change, uint32_t* depth)
{
    std::vector<AssembledAllele> out;
    *depth = 0;
    if (K == 0 || K > 15 || slice.size() < 2 * (size_t)K) return out;
    const uint32_t mask = K == 16 ? ~0u : ((1u << (2 * K)) - 1);
    std::unordered_map<uint32_t, uint32_t> count;
    for (const std::string& r : pile) {
        uint32_t v = 0, run = 0;
        for (size_t p = 0; p < r.size(); ++p) {
            const int c = nt4((unsigned char)r[p]);
            v = ((v << 2) | (uint32_t)(c & 3)) & mask;
            run = c > 3 ? 0 : run + 1;
            if (run >= K) ++count[v];
        }
    }
    auto code_at = [&](const std::string& t, size_t i, uint32_t& v) -> bool {
        uint64_t w;
        if (i + K > t.size() || !pack_kmer(t.data() + i, K, w)) return false;
        v = (uint32_t)w;
        return true;
    };
    auto cnt = [&](uint32_t v) -> uint32_t {
        const auto it = count.find(v);
        return it == count.end() ? 0u : it->second;
    };
    std::unordered_map<uint32_t, char> on_consensus; // the slice's own k-mers
    for (size_t i = 0; i + K <= slice.size(); ++i) {
        uint32_t v;
        if (code_at(slice, i, v)) on_consensus[v] = 1;
    }
    const size_t n_starts = flank_l >= K ? (size_t)(flank_l - K + 1) : 0, n_ends = flank_r >= K ? (size_t)(flank_r - K + 1) : 0;
    for (size_t i = 0; i < n_starts; ++i) {
        uint32_t v;
        if (code_at(slice, i, v)) *depth = std::max(*depth, cnt(v));
    }
    for (size_t i = 0; i < n_ends; ++i) {
        uint32_t v;
        if (code_at(slice, slice.size() - K - i, v)) *depth = std::max(*depth, cnt(v));
    }
    for (size_t si = 0; si < n_starts && out.empty(); ++si) {
        uint32_t sv;
        if (!code_at(slice, si, sv) || cnt(sv) < DBG_MIN_DP) continue;
        for (size_t ei = 0; ei < n_ends; ++ei) {
            const size_t eo = slice.size() - K - ei; // the end k-mer's offset in the slice
            uint32_t ev;
            if (eo <= si || !code_at(slice, eo, ev) || cnt(ev) < DBG_MIN_DP) continue;
            const size_t want = eo + K - si, max_len = want + max_len_change; // bases from the start k-mer's first to the end k-mer's last
            // depth-first over the graph; a path is the bases appended to the start k-mer
            std::vector<std::string> paths;
            std::string cur;
            struct Frame {
                uint32_t v;
                int next; // next base to try
            };
            std::vector<Frame> st { { sv, 0 } };
            uint32_t steps = 0;
            bool too_many = false;
            while (!st.empty() && !too_many) {
                Frame& f = st.back();
                if (f.next == 0 && f.v == ev && K + cur.size() >= (want > max_len_change ? want - max_len_change : K)) {
                    paths.push_back(cur);
                    if (paths.size() > DBG_MAX_PATHS) too_many = true;
                }
                if (f.next > 3 || K + cur.size() >= max_len) {
                    st.pop_back();
                    if (!cur.empty()) cur.pop_back();
                    continue;
                }
                const int b = f.next++;
                const uint32_t nv = ((f.v << 2) | (uint32_t)b) & mask;
                if (++steps > DBG_MAX_STEPS) too_many = true;
                if (cnt(nv) < DBG_MIN_DP) continue;
                cur.push_back("ACGT"[b]);
                st.push_back({ nv, 0 });
            }
            if (too_many) continue; // (a repeat: this pair of k-mers does not delimit the region)
            std::map<std::string, uint32_t> found;
            for (const std::string& tail : paths) {
                const std::string spelled = slice.substr(si, K) + tail;
                std::string whole = slice.substr(0, si) + spelled + slice.substr(eo + K);
                if (whole == slice) continue;
                uint32_t support = ~0u, v;
                for (size_t i = 0; i + K <= spelled.size(); ++i)
                    if (code_at(spelled, i, v) && !on_consensus.count(v)) support = std::min(support, cnt(v));
                if (support == ~0u) continue; // (only consensus k-mers in another order: a repeat, not a variant)
                auto it = found.find(whole);
                if (it == found.end() || it->second < support) found[whole] = support;
            }
            if (paths.empty()) continue;
            for (auto& kv : found) out.push_back({ kv.first, kv.second });
            break; // the first pair of k-mers that is connected decides (pandora does the same [UPSTREAM-MEMORY])
        }
    }
    std::sort(out.begin(), out.end(), [](const AssembledAllele& a, const AssembledAllele& b) { return a.support != b.support ? a.support > b.support : a.slice < b.slice; });
    return out;
}

} // namespace

std::vector<NovelVariant> assemble_candidate_regions(const GenotypeResult& gr, const std::string& reads_path, int threads, const DiscoverParams& dp,
    bool accurate_reads, const ResidentReads& resident)
{
    std::vector<NovelVariant> out;
    const uint32_t A = dp.anchor_len;
    if (gr.candidates.empty() || A == 0 || A > 31) return out;
    // anchor k-mers of every region that has both (a region at the very end of a locus has no room for one: left alone).
    // Accurate reads: the one anchor on each side; noisy reads: up to three on each side, because a read with 5 % errors holds a
    // given 15-mer intact only half of the time
    const uint32_t max_anchors = accurate_reads ? 1 : 3;
    std::unordered_multimap<uint64_t, AnchorRef> anchors;
    for (uint32_t r = 0; r < gr.candidates.size(); ++r) {
        const CandidateRegion& c = gr.candidates[r];
        if (c.left_anchor.size() != A || c.right_anchor.size() != A) continue;
        const uint32_t nl = std::min<uint32_t>(max_anchors, (uint32_t)c.left_context.size() / A), nr = std::min<uint32_t>(max_anchors, (uint32_t)c.right_context.size() / A);
        for (uint32_t side = 0; side < 2; ++side)
            for (uint32_t j = 0; j < (side ? nr : nl); ++j) {
                const std::string fwd = side ? c.right_context.substr((size_t)A * j, A) : c.left_context.substr(c.left_context.size() - (size_t)A * (j + 1), A);
                const std::string rev = revcomp(fwd);
                uint64_t f, rc;
                if (!pack_kmer(fwd.data(), A, f) || !pack_kmer(rev.data(), A, rc)) continue;
                anchors.insert({ f, AnchorRef { r, (uint8_t)side, 0, (uint8_t)j } });
                anchors.insert({ rc, AnchorRef { r, (uint8_t)side, 1, (uint8_t)j } });
            }
    }
    if (anchors.empty()) return out;
    // local assembly (accurate reads): every k-mer of a region's slice of the consensus, either orientation -> (region, orientation); a read
    // that holds one joins the region's pile
    enum { MODE_PILEUP = 1, MODE_DBG = 2 };
    int mode = accurate_reads ? (MODE_PILEUP | MODE_DBG) : MODE_PILEUP;
    if (const char* e = std::getenv("DRPRG_HIP_DENOVO")) {
        if (!std::strcmp(e, "pileup")) mode = MODE_PILEUP;
        else if (!std::strcmp(e, "dbg") && accurate_reads) mode = MODE_DBG;
    }
    struct SliceRef {
        uint32_t region;
        uint8_t reverse;
    };
    std::unordered_multimap<uint64_t, SliceRef> slice_kmers;
    std::vector<std::string> slices(gr.candidates.size());
    if (mode & MODE_DBG)
        for (uint32_t r = 0; r < gr.candidates.size(); ++r) {
            const CandidateRegion& c = gr.candidates[r];
            if (c.left_anchor.size() != A || c.right_anchor.size() != A) continue;
            slices[r] = c.left_anchor + c.seq + c.right_anchor;
            const std::string rc_slice = revcomp(slices[r]);
            for (int rev = 0; rev < 2; ++rev) {
                const std::string& t = rev ? rc_slice : slices[r];
                for (size_t i = 0; i + A <= t.size(); ++i) {
                    uint64_t v;
                    if (pack_kmer(t.data() + i, A, v)) slice_kmers.insert({ v, SliceRef { r, (uint8_t)rev } });
                }
            }
        }
    struct RegionPile {
        std::mutex mu;
        std::vector<std::string> reads;
    };
    std::vector<RegionPile> piles(gr.candidates.size());
    // low 20 bits of the packed k-mer, one BIT each (128 KB: stays in the L2 of every parser thread; with the slices' k-mers in it -- a
    // few hundred per region -- the 8 KB table of round 3 let one read k-mer in eight through to the hash maps): most read k-mers stop here
    std::vector<uint64_t> prefilter(1u << 14, 0);
    for (auto& kv : anchors) prefilter[(kv.first & 0xFFFFF) >> 6] |= 1ull << (kv.first & 63);
    for (auto& kv : slice_kmers) prefilter[(kv.first & 0xFFFFF) >> 6] |= 1ull << (kv.first & 63);
    uint8_t code_of[256]; // the scan below is the whole cost of this pass (every base of every read): table, no branches per base
    for (int ch = 0; ch < 256; ++ch) code_of[ch] = (uint8_t)nt4((unsigned char)ch);
    std::vector<RegionVotes> votes(gr.candidates.size());
    const uint64_t mask = A == 32 ? ~0ull : ((1ull << (2 * A)) - 1);

    struct Hit {
        uint32_t region, pos;
        uint8_t right, reverse, j;
    };
    struct Span { // the best pair of anchors of one (region, orientation) in one read: the innermost ones
        uint32_t region, from, len;
        uint8_t reverse, jl, jr;
    };
    auto scan_batch = [&](const PinnedBatch& b) {
        std::vector<Hit> hits;
        std::vector<Span> spans;
        std::vector<SliceRef> touched;
        for (uint64_t i = 0; i < b.n_reads; ++i) {
            const char* s = (const char*)b.bases + b.offsets[i];
            const uint64_t len = b.offsets[i + 1] - b.offsets[i];
            if (len < (uint64_t)A || (len < 2 * (uint64_t)A && !(mode & MODE_DBG))) continue;
            hits.clear();
            touched.clear();
            uint64_t v = 0;
            uint32_t run = 0;
            for (uint64_t p = 0; p < len; ++p) {
                const uint32_t c = code_of[(unsigned char)s[p]];
                v = ((v << 2) | (uint64_t)(c & 3)) & mask;
                run = c > 3 ? 0 : run + 1; // (a k-mer over a non-ACGT base never counts: the run restarts after it)
                if (!((prefilter[(v & 0xFFFFF) >> 6] >> (v & 63)) & 1) || run < A) continue;
                auto range = anchors.equal_range(v);
                for (auto it = range.first; it != range.second; ++it)
                    hits.push_back(Hit { it->second.region, (uint32_t)(p + 1 - A), it->second.right, it->second.reverse, it->second.j });
                if (mode & MODE_DBG) {
                    auto sr = slice_kmers.equal_range(v);
                    for (auto it = sr.first; it != sr.second; ++it) {
                        bool seen = false;
                        for (const SliceRef& t : touched) seen |= t.region == it->second.region && t.reverse == it->second.reverse;
                        if (!seen) touched.push_back(it->second);
                    }
                }
            }
            for (const SliceRef& t : touched) { // the read joins the region's pile, oriented like the consensus
                std::string r(s, len);
                for (char& ch : r) ch = (char)std::toupper((unsigned char)ch);
                if (t.reverse) r = revcomp(r);
                std::lock_guard<std::mutex> g(piles[t.region].mu);
                piles[t.region].reads.push_back(std::move(r));
            }
            if (hits.size() < 2 || !(mode & MODE_PILEUP)) continue;
            // forward read: left anchor, allele, right anchor; reverse read: rc(right anchor), rc(allele), rc(left anchor)
            spans.clear();
            for (const Hit& x : hits)
                for (const Hit& y : hits) {
                    if (x.region != y.region || x.reverse != y.reverse) continue;
                    const bool first_is_x = x.reverse ? (x.right == 1 && y.right == 0) : (x.right == 0 && y.right == 1);
                    if (!first_is_x || y.pos < x.pos + A) continue;
                    const CandidateRegion& c = gr.candidates[x.region];
                    const uint8_t jl = x.reverse ? y.j : x.j, jr = x.reverse ? x.j : y.j;
                    const uint32_t got = y.pos - (x.pos + A), want = c.end - c.start + A * ((uint32_t)jl + jr);
                    if (got > want + dp.max_len_change || got + dp.max_len_change < want) continue;
                    Span sp { x.region, x.pos + A, got, x.reverse, jl, jr };
                    bool placed = false;
                    for (Span& o : spans)
                        if (o.region == sp.region && o.reverse == sp.reverse) {
                            if ((uint32_t)sp.jl + sp.jr < (uint32_t)o.jl + o.jr) o = sp;
                            placed = true;
                        }
                    if (!placed) spans.push_back(sp);
                }
            for (const Span& sp : spans) {
                std::string allele(s + sp.from, sp.len);
                for (char& ch : allele) ch = (char)std::toupper((unsigned char)ch);
                if (sp.reverse) allele = revcomp(allele);
                if (allele.find('N') != std::string::npos) continue;
                const char key[3] = { (char)('0' + sp.jl), (char)('0' + sp.jr), 0 };
                std::lock_guard<std::mutex> g(votes[sp.region].mu);
                votes[sp.region].alleles[key + allele] += 1;
            }
        }
    };
    if (resident) {
        // the reads are in HBM: the device picks those that hold an anchor, the scan above runs on them alone
        std::vector<uint64_t> kmers;
        for (auto& kv : anchors) kmers.push_back(kv.first);
        for (auto& kv : slice_kmers) kmers.push_back(kv.first);
        std::sort(kmers.begin(), kmers.end());
        kmers.erase(std::unique(kmers.begin(), kmers.end()), kmers.end());
        std::vector<uint8_t> sel_bases;
        std::vector<uint64_t> sel_offsets;
        resident(kmers, A, sel_bases, sel_offsets);
        if (sel_offsets.size() > 1) {
            sel_bases.resize(sel_bases.size() + 64); // (what an ingest block has after its last base)
            PinnedBatch b { sel_bases.data(), sel_offsets.data(), sel_offsets.size() - 1, sel_offsets.back() };
            scan_batch(b);
        }
    } else {
        IngestHooks hooks;
        hooks.concurrent_submit = true;
        hooks.submit = scan_batch;
        try {
            ingest_fastx(reads_path, threads, hooks);
        } catch (const Error& e) {
            if (e.code != DRPRG_EAGAIN_SERIAL) throw;
            // multi-line FASTQ: the serial reader (fastx.cpp), batch by batch through the same scan
            for (RegionVotes& v : votes) v.alleles.clear();
            for (RegionPile& pl : piles) pl.reads.clear();
            FastxReader reader(reads_path);
            ReadBatch rb;
            while (reader.next_batch(rb, 1u << 18, 64ull << 20)) {
                if (!rb.n_reads()) continue;
                rb.bases.resize(rb.bases.size() + 64); // (what an ingest block has after its last base)
                PinnedBatch b { rb.bases.data(), rb.offsets.data(), rb.n_reads(), rb.offsets.back() };
                scan_batch(b);
            }
        }
    }
    // a variant from two spellings of the same stretch of the consensus (which starts at consensus position `origin`)
    auto make_variant = [&](const CandidateRegion& c, uint32_t origin, const std::string& ref, const std::string& alt, uint32_t support, uint32_t spanning) {
        size_t pre = 0;
        while (pre < ref.size() && pre < alt.size() && ref[pre] == alt[pre]) ++pre;
        size_t suf = 0;
        while (suf < ref.size() - pre && suf < alt.size() - pre && ref[ref.size() - 1 - suf] == alt[alt.size() - 1 - suf]) ++suf;
        NovelVariant v;
        v.chrom = c.chrom;
        v.prg = c.prg;
        v.pos = origin + (uint32_t)pre;
        v.ref = ref.substr(pre, ref.size() - pre - suf);
        v.alt = alt.substr(pre, alt.size() - pre - suf);
        v.support = support;
        v.spanning = spanning;
        v.group = (uint32_t)(&c - gr.candidates.data());
        return v;
    };
    for (uint32_t r = 0; r < gr.candidates.size(); ++r) {
        const CandidateRegion& c = gr.candidates[r];
        const size_t first_of_region = out.size();
        uint32_t spanning = 0, best_n = 0;
        const std::string* best_key = nullptr;
        for (auto& kv : votes[r].alleles) {
            spanning += kv.second;
            if (kv.second > best_n || (kv.second == best_n && best_key && kv.first < *best_key)) {
                best_n = kv.second;
                best_key = &kv.first;
            }
        }
        std::string allele = c.seq;
        if (!accurate_reads) { // noisy reads: the allele is the column-wise majority of the aligned strings
            if (spanning >= dp.min_support) {
                // a change the first alignment spreads over neighbouring columns (a deletion inside a repeat) is gathered by aligning again,
                // to the result: at most three rounds, normally the second one changes nothing
                for (int round = 0; round < 3; ++round) {
                    std::string next = column_consensus(c, allele, A, votes[r].alleles, spanning, dp);
                    if (next == allele) break;
                    allele.swap(next);
                }
                if (allele != c.seq) {
                    best_n = 0;
                    for (auto& kv : votes[r].alleles) { // reported support: reads that are closer to the new allele than to the consensus
                        const uint32_t jl = (uint32_t)(kv.first[0] - '0'), jr = (uint32_t)(kv.first[1] - '0');
                        const std::string s = kv.first.substr(2);
                        best_n += kv.second * (edit_distance(s, extended(c, A, jl, jr, allele)) < edit_distance(s, extended(c, A, jl, jr, c.seq)));
                    }
                }
            }
        } else if ((mode & MODE_PILEUP) && best_key && best_n >= dp.min_support && (double)best_n >= dp.min_fraction * (double)spanning) {
            allele = best_key->substr(2); // one anchor per side: the key is "00" + the string
        }
        if (allele != c.seq) out.push_back(make_variant(c, c.start, c.seq, allele, best_n, spanning));
        // local assembly: every assembled allele with enough support that the pile-up has not reported already
        if ((mode & MODE_DBG) && !slices[r].empty()) {
            uint32_t depth = 0;
            const uint32_t flank_l = A + (c.low_start - c.start), flank_r = A + (c.end - c.low_end);
            const std::vector<AssembledAllele> found = assemble_region(slices[r], flank_l, flank_r, piles[r].reads, A, dp.max_len_change, &depth);
            const uint32_t need = found.empty() ? 0 : std::max<uint32_t>(dp.min_support, (found[0].support + 9) / 10); // a tenth of the best one
            for (const AssembledAllele& f : found) {
                if (f.support < need) break;
                NovelVariant v = make_variant(c, c.start - A, slices[r], f.slice, f.support, depth);
                bool dup = false;
                for (size_t i = first_of_region; i < out.size(); ++i) dup |= out[i].pos == v.pos && out[i].ref == v.ref && out[i].alt == v.alt;
                if (!dup) out.push_back(std::move(v));
            }
        }
    }
    std::sort(out.begin(), out.end(), [](const NovelVariant& a, const NovelVariant& b) {
        if (a.chrom != b.chrom) return a.chrom < b.chrom;
        return a.pos < b.pos;
    });
    return out;
}

void write_denovo_paths(const std::string& dir, const std::string& sample, const GenotypeResult& gr, const std::vector<NovelVariant>& variants,
    bool list_loci)
{
    std::map<std::string, std::vector<const NovelVariant*>> by_locus;
    for (const NovelVariant& v : variants) by_locus[v.chrom].push_back(&v);
    {
        std::ofstream o(dir + "/denovo_variants.tsv");
        o << "#locus\tpos\tref\talt\treads_with_alt\treads_spanning\n";
        for (const NovelVariant& v : variants)
            o << v.chrom << "\t" << v.pos + 1 << "\t" << (v.ref.empty() ? "." : v.ref) << "\t" << (v.alt.empty() ? "." : v.alt) << "\t" << v.support << "\t"
              << v.spanning << "\n";
        if (!o) throw Error(DRPRG_EIO, "cannot write " + dir + "/denovo_variants.tsv");
    }
    std::ofstream o(dir + "/denovo_paths.txt"), fa(dir + "/denovo_sequences.fa");
    o << "1 samples\nSample " << sample << "\n" << (list_loci ? by_locus.size() : 0) << " loci with denovo variants\n";
    if (list_loci)
        for (auto& kv : by_locus) {
            const LocusConsensus* lc = nullptr;
            for (const LocusConsensus& c : gr.consensus)
                if (c.chrom == kv.first) lc = &c;
            if (!lc) continue;
            o << kv.first << "\n" << lc->nodes.size() << " nodes\n";
            for (const ConsensusNode& n : lc->nodes) o << "(" << n.id << " [" << n.start << ", " << n.end << ") " << n.seq << ")\n";
            o << kv.second.size() << " denovo variants for this locus\n";
            // the locus' sequence with the variants applied; of variants that overlap (alleles of one region) the best supported one
            std::string updated = lc->seq;
            std::vector<const NovelVariant*> by_support = kv.second, chosen;
            std::stable_sort(by_support.begin(), by_support.end(), [](const NovelVariant* x, const NovelVariant* y) { return x->support > y->support; });
            for (const NovelVariant* v : by_support) {
                bool clash = false;
                for (const NovelVariant* u : chosen)
                    clash |= (v->pos <= u->pos + u->ref.size() && u->pos <= v->pos + v->ref.size()) || (u->group != ~0u && u->group == v->group);
                if (!clash) chosen.push_back(v);
            }
            std::sort(chosen.begin(), chosen.end(), [](const NovelVariant* x, const NovelVariant* y) { return x->pos > y->pos; });
            for (const NovelVariant* v : chosen) updated.replace(v->pos, v->ref.size(), v->alt); // (right to left: earlier positions stay valid)
            for (const NovelVariant* v : kv.second) o << v->pos + 1 << "\t" << v->ref << "\t" << v->alt << "\n";
            fa << ">" << kv.first << "\n" << updated << "\n";
        }
    if (!o) throw Error(DRPRG_EIO, "cannot write " + dir + "/denovo_paths.txt");
}

// ---- PRG update ---------------------------------------------------------------------------------------------------------------------
// The reference adds the consensus-with-variants of a locus to the locus' multiple alignment (mafft --add) and lets make_prg build the PRG
// again (/root/reference/src/lib.rs:279-456); both tools are external.  Here the PRG string itself is edited so that its language grows by
// exactly the new sequence: the stretch of the string the variant touches -- widened until it holds only whole sites (a variant that runs
// into or across a site takes the whole site) -- becomes the first allele of a new site whose second allele is what the called path spells
// there with the variant applied:   ... <M> old stretch <M+1> path with variant <M> ...   Stretches of several variants that overlap are
// joined first.  A variant inside one node gives the plain  <M> ref <M+1> alt <M>  of rounds 2-3.  Afterwards the markers are numbered
// again in the order pandora's parser meets them (a site before the sites inside it, then the sites behind it; 5, 7, 9 ...): its
// LocalPRG::build_graph splits an interval at the NEXT marker number and stops if another site comes first [UPSTREAM-MEMORY], which is
// also the numbering of the reference's own dr.prg (" 21  23 G 24 T 23").
namespace {

struct PrgSite {
    size_t open_off = 0, close_off = 0; // [the space before the opening marker, one past the space behind the closing marker)
    int parent_ctx = 0;
};
struct PrgCtx {
    int parent = -1, site = -1, depth = 0;
};
// context (site, allele) of every base of the string; markers are " <n> " with odd n opening and closing a site and n + 1 between alleles
bool scan_prg(const std::string& prg, std::vector<PrgSite>& sites, std::vector<PrgCtx>& ctxs, std::vector<int>& ctx_of)
{
    sites.clear();
    ctxs.assign(1, PrgCtx {});
    ctx_of.assign(prg.size(), -1);
    struct Open {
        int marker, site;
    };
    std::vector<Open> stack;
    int cur = 0;
    for (size_t i = 0; i < prg.size();) {
        const unsigned char c = (unsigned char)prg[i];
        if (std::isdigit(c)) {
            size_t j = i;
            int m = 0;
            while (j < prg.size() && std::isdigit((unsigned char)prg[j])) m = m * 10 + (prg[j++] - '0');
            if (m % 2) {
                if (!stack.empty() && stack.back().marker == m) { // closes
                    sites[(size_t)stack.back().site].close_off = std::min(j + 1, prg.size());
                    cur = sites[(size_t)stack.back().site].parent_ctx;
                    stack.pop_back();
                } else { // opens
                    PrgSite st;
                    st.open_off = i ? i - 1 : 0;
                    st.parent_ctx = cur;
                    sites.push_back(st);
                    stack.push_back({ m, (int)sites.size() - 1 });
                    ctxs.push_back({ cur, (int)sites.size() - 1, ctxs[(size_t)cur].depth + 1 });
                    cur = (int)ctxs.size() - 1;
                }
            } else {
                if (stack.empty() || stack.back().marker + 1 != m) return false;
                const int st = stack.back().site;
                ctxs.push_back({ sites[(size_t)st].parent_ctx, st, ctxs[(size_t)sites[(size_t)st].parent_ctx].depth + 1 });
                cur = (int)ctxs.size() - 1;
            }
            i = j;
        } else {
            if (c != ' ') ctx_of[i] = cur;
            ++i;
        }
    }
    return stack.empty();
}

// markers in the order pandora's parser consumes them: pre-order, left to right
std::string renumber_prg(const std::string& prg)
{
    std::string out;
    out.reserve(prg.size() + 16);
    std::vector<std::pair<int, int>> stack; // (old marker, new marker)
    int next = 5;
    for (size_t i = 0; i < prg.size();) {
        if (std::isdigit((unsigned char)prg[i])) {
            size_t j = i;
            int m = 0;
            while (j < prg.size() && std::isdigit((unsigned char)prg[j])) m = m * 10 + (prg[j++] - '0');
            int nm;
            if (m % 2) {
                if (!stack.empty() && stack.back().first == m) {
                    nm = stack.back().second;
                    stack.pop_back();
                } else {
                    nm = next;
                    next += 2;
                    stack.push_back({ m, nm });
                }
            } else
                nm = stack.empty() ? m : stack.back().second + 1;
            out += std::to_string(nm);
            i = j;
        } else
            out += prg[i++];
    }
    return out;
}

} // namespace

// Returns the number of variants applied; `skipped`: variants that could not be placed (a locus without a called path, a position
// outside it, a PRG string that does not scan).
uint32_t update_prgs(std::vector<std::pair<std::string, std::string>>& prgs, const GenotypeResult& gr, const std::vector<NovelVariant>& variants,
    std::vector<std::string>* skipped)
{
    uint32_t applied = 0;
    std::map<uint32_t, std::vector<const NovelVariant*>> by_prg;
    for (const NovelVariant& v : variants) by_prg[v.prg].push_back(&v);
    auto skip = [&](const NovelVariant* v) {
        if (skipped) skipped->push_back(v->chrom + ":" + std::to_string(v->pos + 1));
    };
    for (auto& kv : by_prg) {
        const LocusConsensus* lc = nullptr;
        for (const LocusConsensus& c : gr.consensus)
            if (c.prg == kv.first) lc = &c;
        std::vector<PrgSite> sites;
        std::vector<PrgCtx> ctxs;
        std::vector<int> ctx_of;
        if (kv.first >= prgs.size() || !lc || !scan_prg(prgs[kv.first].second, sites, ctxs, ctx_of)) {
            for (const NovelVariant* v : kv.second) skip(v);
            continue;
        }
        std::string& prg = prgs[kv.first].second;
        const std::string& cons = lc->seq;
        // consensus position -> offset in the PRG string
        std::vector<uint32_t> cum(lc->nodes.size() + 1, 0);
        for (size_t i = 0; i < lc->nodes.size(); ++i) cum[i + 1] = cum[i] + (lc->nodes[i].end - lc->nodes[i].start);
        auto offset_of = [&](uint32_t p) -> size_t {
            size_t i = (size_t)(std::upper_bound(cum.begin(), cum.end(), p) - cum.begin()) - 1; // the node that holds base p (empty nodes skipped)
            return (size_t)lc->nodes[i].start + (p - cum[i]);
        };
        auto first_path_base_at_or_after = [&](size_t off) -> uint32_t {
            for (size_t i = 0; i < lc->nodes.size(); ++i)
                if (lc->nodes[i].start >= off && lc->nodes[i].end > lc->nodes[i].start) return cum[i];
            return cum.back();
        };
        struct Edit {
            size_t a, b;     // stretch of the PRG string
            uint32_t p, q;   // the path bases inside it
            std::vector<const NovelVariant*> vs;
        };
        std::vector<Edit> edits;
        // the alleles of one candidate region (same group) share ONE stretch: the span from the first to the last of them
        std::map<uint32_t, std::vector<const NovelVariant*>> units;
        uint32_t solo = 0;
        for (const NovelVariant* v : kv.second) units[v->group != ~0u ? v->group : 0x80000000u + solo++].push_back(v);
        for (auto& unit : units) {
            const std::vector<const NovelVariant*>& members = unit.second;
            const NovelVariant* v = members[0];
            uint32_t p = v->pos, q = v->pos + (uint32_t)v->ref.size();
            for (const NovelVariant* m : members) {
                p = std::min(p, m->pos);
                q = std::max(q, m->pos + (uint32_t)m->ref.size());
            }
            if (q > cons.size() || cum.back() != cons.size() || cons.empty()) {
                for (const NovelVariant* m : members) skip(m);
                continue;
            }
            if (p == q) { // an insertion takes the base before it along (the one behind it at the very start)
                if (p) --p;
                else ++q;
            }
            const size_t xa = offset_of(p), xb = offset_of(q - 1);
            if (xa >= prg.size() || xb >= prg.size() || ctx_of[xa] < 0 || ctx_of[xb] < 0) {
                for (const NovelVariant* m : members) skip(m);
                continue;
            }
            // the deepest context both ends lie in; an end that lies deeper takes the whole site it is in at the next level
            int ca = ctx_of[xa], cb = ctx_of[xb], sa = -1, sb = -1;
            while (ctxs[(size_t)ca].depth > ctxs[(size_t)cb].depth) sa = ctxs[(size_t)ca].site, ca = ctxs[(size_t)ca].parent;
            while (ctxs[(size_t)cb].depth > ctxs[(size_t)ca].depth) sb = ctxs[(size_t)cb].site, cb = ctxs[(size_t)cb].parent;
            while (ca != cb) {
                sa = ctxs[(size_t)ca].site, ca = ctxs[(size_t)ca].parent;
                sb = ctxs[(size_t)cb].site, cb = ctxs[(size_t)cb].parent;
            }
            Edit e;
            e.a = sa >= 0 ? sites[(size_t)sa].open_off : xa;
            e.b = sb >= 0 ? sites[(size_t)sb].close_off : xb + 1;
            e.p = sa >= 0 ? first_path_base_at_or_after(e.a) : p;
            e.q = sb >= 0 ? first_path_base_at_or_after(e.b) : q;
            e.vs = members;
            edits.push_back(std::move(e));
        }
        // overlapping stretches lie in one context (each holds whole sites only), so their union is a stretch of the same kind
        std::sort(edits.begin(), edits.end(), [](const Edit& x, const Edit& y) { return x.a < y.a; });
        std::vector<Edit> merged;
        for (Edit& e : edits) {
            if (!merged.empty() && e.a < merged.back().b) {
                Edit& m = merged.back();
                m.b = std::max(m.b, e.b);
                m.p = std::min(m.p, e.p);
                m.q = std::max(m.q, e.q);
                m.vs.insert(m.vs.end(), e.vs.begin(), e.vs.end());
            } else
                merged.push_back(std::move(e));
        }
        int marker = 3;
        for (size_t i = 0; i < prg.size();) {
            if (std::isdigit((unsigned char)prg[i])) {
                int v = 0;
                while (i < prg.size() && std::isdigit((unsigned char)prg[i])) v = v * 10 + (prg[i++] - '0');
                marker = std::max(marker, v);
            } else ++i;
        }
        int next_marker = marker + 1 + ((marker + 1) % 2 == 0 ? 1 : 0); // the next odd number (renumbered below)
        for (auto it = merged.rbegin(); it != merged.rend(); ++it) { // right to left: the offsets further left stay valid
            const std::string base = cons.substr(it->p, it->q - it->p);
            std::vector<const NovelVariant*> vs = it->vs;
            std::sort(vs.begin(), vs.end(), [](const NovelVariant* x, const NovelVariant* y) { return x->pos > y->pos; });
            // Variants of one candidate region (same group) and variants that overlap each other (the overlapping lines of a
            // denovo_paths.txt) are alternatives: each becomes an allele of its own; the others are in every allele.
            auto overlap = [](const NovelVariant* x, const NovelVariant* y) {
                const uint32_t xe = x->pos + (uint32_t)std::max<size_t>(x->ref.size(), 1), ye = y->pos + (uint32_t)std::max<size_t>(y->ref.size(), 1);
                return x->pos < ye && y->pos < xe;
            };
            std::vector<const NovelVariant*> common, alternatives;
            bool ok = true;
            for (const NovelVariant* v : vs) {
                if (v->pos < it->p || v->pos + v->ref.size() > it->q) ok = false;
                bool clashes = false;
                for (const NovelVariant* u : vs) clashes |= u != v && (overlap(u, v) || (u->group != ~0u && u->group == v->group));
                (clashes ? alternatives : common).push_back(v);
            }
            if (!ok) {
                for (const NovelVariant* v : vs) skip(v);
                continue;
            }
            auto spelled = [&](const NovelVariant* extra) { // `common` (+ one alternative) applied right to left
                std::vector<const NovelVariant*> use = common;
                if (extra) use.push_back(extra);
                std::sort(use.begin(), use.end(), [](const NovelVariant* x, const NovelVariant* y) { return x->pos > y->pos; });
                std::string t = base;
                for (const NovelVariant* v : use) t.replace(v->pos - it->p, v->ref.size(), v->alt);
                return t;
            };
            std::vector<std::string> alts;
            if (alternatives.empty()) alts.push_back(spelled(nullptr));
            for (const NovelVariant* v : alternatives) {
                const std::string t = spelled(v);
                if (std::find(alts.begin(), alts.end(), t) == alts.end()) alts.push_back(t);
            }
            const std::string m = std::to_string(next_marker), sep = std::to_string(next_marker + 1);
            std::string site = " " + m + " " + prg.substr(it->a, it->b - it->a);
            for (const std::string& t : alts) site += " " + sep + " " + t;
            prg = prg.substr(0, it->a) + site + " " + m + " " + prg.substr(it->b);
            next_marker += 2;
            applied += (uint32_t)vs.size();
        }
        prg = renumber_prg(prg);
    }
    return applied;
}

void read_denovo_paths(const std::string& path, const std::vector<std::string>& names, GenotypeResult& gr, std::vector<NovelVariant>& variants)
{
    std::ifstream in(path);
    if (!in) throw Error(DRPRG_EIO, "cannot open " + path);
    auto bad = [&](const std::string& what) -> Error { return Error(DRPRG_EINVAL, path + ": " + what); };
    std::string line;
    auto next_line = [&]() -> bool {
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (!line.empty()) return true;
        }
        return false;
    };
    auto leading_count = [&](const std::string& l, const char* tail) -> long { // "<n> <tail>..." -> n, else -1
        size_t i = 0;
        while (i < l.size() && std::isdigit((unsigned char)l[i])) ++i;
        if (!i || l.compare(i, std::strlen(tail), tail) != 0) return -1;
        return std::stol(l.substr(0, i));
    };
    if (!next_line() || leading_count(line, " samples") < 0) throw bad("no '<n> samples' line");
    if (!next_line() || line.compare(0, 7, "Sample ") != 0) throw bad("no 'Sample <name>' line");
    if (!next_line()) throw bad("no '<n> loci with denovo variants' line");
    const long n_loci = leading_count(line, " loci with denovo variants");
    if (n_loci < 0) throw bad("no '<n> loci with denovo variants' line");
    for (long l = 0; l < n_loci; ++l) {
        if (!next_line()) throw bad("fewer loci than announced");
        LocusConsensus lc;
        lc.chrom = line;
        const auto it = std::find(names.begin(), names.end(), lc.chrom);
        if (it == names.end()) throw bad("locus " + lc.chrom + " is not in the PRG file");
        lc.prg = (uint32_t)(it - names.begin());
        if (!next_line()) throw bad("no '<n> nodes' line for " + lc.chrom);
        const long n_nodes = leading_count(line, " nodes");
        if (n_nodes < 0) throw bad("no '<n> nodes' line for " + lc.chrom);
        for (long i = 0; i < n_nodes; ++i) {
            if (!next_line()) throw bad("fewer nodes than announced for " + lc.chrom);
            // (id [start, end) seq)
            ConsensusNode n;
            unsigned id = 0, a = 0, b = 0;
            int used = 0;
            if (std::sscanf(line.c_str(), "(%u [%u, %u) %n", &id, &a, &b, &used) < 3 || used <= 0 || line.back() != ')') throw bad("node line '" + line + "'");
            n.id = id;
            n.start = a;
            n.end = b;
            n.seq = line.substr((size_t)used, line.size() - 1 - (size_t)used);
            if (n.end < n.start || n.seq.size() != n.end - n.start) throw bad("node interval and sequence disagree in '" + line + "'");
            lc.seq += n.seq;
            lc.nodes.push_back(std::move(n));
        }
        if (!next_line()) throw bad("no '<n> denovo variants for this locus' line for " + lc.chrom);
        const long n_var = leading_count(line, " denovo variants for this locus");
        if (n_var < 0) throw bad("no '<n> denovo variants for this locus' line for " + lc.chrom);
        for (long i = 0; i < n_var; ++i) {
            if (!next_line()) throw bad("fewer variants than announced for " + lc.chrom);
            const size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1);
            if (t2 == std::string::npos) throw bad("variant line '" + line + "'");
            NovelVariant v;
            v.chrom = lc.chrom;
            v.prg = lc.prg;
            const long pos1 = std::atol(line.substr(0, t1).c_str());
            v.ref = line.substr(t1 + 1, t2 - t1 - 1);
            v.alt = line.substr(t2 + 1);
            if (pos1 < 1 || (size_t)pos1 - 1 + v.ref.size() > lc.seq.size() || lc.seq.compare((size_t)pos1 - 1, v.ref.size(), v.ref) != 0)
                throw bad("variant '" + line + "' does not lie on the path of " + lc.chrom);
            v.pos = (uint32_t)(pos1 - 1);
            variants.push_back(std::move(v));
        }
        gr.consensus.push_back(std::move(lc));
    }
}

} // namespace drprg
