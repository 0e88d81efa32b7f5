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
// a novel variant.  Noisy long reads need a real multiple alignment and are left alone (regions are still reported).
//
// Outputs: denovo_paths.txt in the layout the reference's parser and make_prg read (/root/reference/src/lib.rs:648-697 and
// the example at :3010-3038: locus, "<n> nodes", one "(id [start, end) seq)" line per local node of the called path,
// "<m> denovo variants for this locus", "pos<TAB>ref<TAB>alt" lines -- positions 1-based on the path sequence
// [UPSTREAM-MEMORY]); and, for hosts without make_prg / mafft, the PRG updated in place: a variant that lies inside ONE local
// node becomes a new site of that PRG (MakePrg::update's job in the reference, /root/reference/src/lib.rs:279-456).
#include "denovo.h"
#include "ingest.h"
#include <algorithm>
#include <atomic>
#include <fstream>
#include <mutex>
#include <sstream>
#include <unordered_map>

namespace drprg {

namespace {

struct AnchorRef {
    uint32_t region;
    uint8_t right;   // 0 = left anchor, 1 = right anchor
    uint8_t reverse; // the reverse complement of the anchor (the read runs against the consensus)
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

} // namespace

std::vector<NovelVariant> assemble_candidate_regions(const GenotypeResult& gr, const std::string& reads_path, int threads, const DiscoverParams& dp)
{
    std::vector<NovelVariant> out;
    const uint32_t A = dp.anchor_len;
    if (gr.candidates.empty() || A == 0 || A > 31) return out;
    // anchor k-mers of every region that has both (a region at the very end of a locus has no room for one: left alone)
    std::unordered_multimap<uint64_t, AnchorRef> anchors;
    for (uint32_t r = 0; r < gr.candidates.size(); ++r) {
        const CandidateRegion& c = gr.candidates[r];
        if (c.left_anchor.size() != A || c.right_anchor.size() != A) continue;
        uint64_t l, rr, lrc, rrc;
        if (!pack_kmer(c.left_anchor.data(), A, l) || !pack_kmer(c.right_anchor.data(), A, rr)) continue;
        const std::string lr = revcomp(c.left_anchor), rrs = revcomp(c.right_anchor);
        if (!pack_kmer(lr.data(), A, lrc) || !pack_kmer(rrs.data(), A, rrc)) continue;
        anchors.insert({ l, AnchorRef { r, 0, 0 } });
        anchors.insert({ rr, AnchorRef { r, 1, 0 } });
        anchors.insert({ lrc, AnchorRef { r, 0, 1 } });
        anchors.insert({ rrc, AnchorRef { r, 1, 1 } });
    }
    if (anchors.empty()) return out;
    std::vector<uint8_t> prefilter(1u << 16, 0); // low 16 bits of the packed k-mer: most read k-mers stop here
    for (auto& kv : anchors) prefilter[kv.first & 0xFFFF] = 1;
    std::vector<RegionVotes> votes(gr.candidates.size());
    const uint64_t mask = A == 32 ? ~0ull : ((1ull << (2 * A)) - 1);

    struct Hit {
        uint32_t region, pos;
        uint8_t right, reverse;
    };
    auto scan_batch = [&](const PinnedBatch& b) {
        std::vector<Hit> hits;
        for (uint64_t i = 0; i < b.n_reads; ++i) {
            const char* s = (const char*)b.bases + b.offsets[i];
            const uint64_t len = b.offsets[i + 1] - b.offsets[i];
            if (len < 2 * (uint64_t)A) continue;
            hits.clear();
            uint64_t v = 0;
            uint32_t run = 0;
            for (uint64_t p = 0; p < len; ++p) {
                const int c = nt4((unsigned char)s[p]);
                if (c > 3) {
                    run = 0;
                    continue;
                }
                v = ((v << 2) | (uint64_t)c) & mask;
                if (++run < A || !prefilter[v & 0xFFFF]) continue;
                auto range = anchors.equal_range(v);
                for (auto it = range.first; it != range.second; ++it)
                    hits.push_back(Hit { it->second.region, (uint32_t)(p + 1 - A), it->second.right, it->second.reverse });
            }
            if (hits.size() < 2) continue;
            // forward read: left anchor, allele, right anchor; reverse read: rc(right anchor), rc(allele), rc(left anchor)
            for (const Hit& x : hits)
                for (const Hit& y : hits) {
                    if (x.region != y.region || x.reverse != y.reverse) continue;
                    const bool first_is_x = x.reverse ? (x.right == 1 && y.right == 0) : (x.right == 0 && y.right == 1);
                    if (!first_is_x || y.pos < x.pos + A) continue;
                    const CandidateRegion& c = gr.candidates[x.region];
                    const uint32_t got = y.pos - (x.pos + A), want = c.end - c.start;
                    if (got > want + dp.max_len_change || got + dp.max_len_change < want) continue;
                    std::string allele(s + x.pos + A, got);
                    for (char& ch : allele) ch = (char)std::toupper((unsigned char)ch);
                    if (x.reverse) allele = revcomp(allele);
                    if (allele.find('N') != std::string::npos) continue;
                    std::lock_guard<std::mutex> g(votes[x.region].mu);
                    votes[x.region].alleles[allele] += 1;
                }
        }
    };
    IngestHooks hooks;
    hooks.concurrent_submit = true;
    hooks.submit = scan_batch;
    try {
        ingest_fastx(reads_path, threads, hooks);
    } catch (const Error& e) {
        if (e.code != DRPRG_EAGAIN_SERIAL) throw;
        throw Error(DRPRG_EFORMAT, "discover: multi-line FASTQ is not supported by the region pile-up");
    }
    for (uint32_t r = 0; r < gr.candidates.size(); ++r) {
        const CandidateRegion& c = gr.candidates[r];
        uint32_t spanning = 0, best_n = 0;
        const std::string* best = nullptr;
        for (auto& kv : votes[r].alleles) {
            spanning += kv.second;
            if (kv.second > best_n || (kv.second == best_n && best && kv.first < *best)) {
                best_n = kv.second;
                best = &kv.first;
            }
        }
        if (!best || *best == c.seq || best_n < dp.min_support || (double)best_n < dp.min_fraction * (double)spanning) continue;
        // trim what the allele shares with the consensus on both sides
        const std::string& ref = c.seq;
        const std::string& alt = *best;
        size_t pre = 0;
        while (pre < ref.size() && pre < alt.size() && ref[pre] == alt[pre]) ++pre;
        size_t suf = 0;
        while (suf < ref.size() - pre && suf < alt.size() - pre && ref[ref.size() - 1 - suf] == alt[alt.size() - 1 - suf]) ++suf;
        NovelVariant v;
        v.chrom = c.chrom;
        v.prg = c.prg;
        v.pos = c.start + (uint32_t)pre;
        v.ref = ref.substr(pre, ref.size() - pre - suf);
        v.alt = alt.substr(pre, alt.size() - pre - suf);
        v.support = best_n;
        v.spanning = spanning;
        out.push_back(std::move(v));
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
            std::string updated = lc->seq;
            for (auto it = kv.second.rbegin(); it != kv.second.rend(); ++it) // (right to left: earlier positions stay valid)
                updated.replace((*it)->pos, (*it)->ref.size(), (*it)->alt);
            for (const NovelVariant* v : kv.second) o << v->pos + 1 << "\t" << v->ref << "\t" << v->alt << "\n";
            fa << ">" << kv.first << "\n" << updated << "\n";
        }
    if (!o) throw Error(DRPRG_EIO, "cannot write " + dir + "/denovo_paths.txt");
}

// A variant that lies inside one local node of the called path (no existing site in the way) becomes a new site of the
// PRG string: "... <M> ref <M+1> alt <M> ..." with M a fresh odd marker.  Returns the number of variants applied.
uint32_t update_prgs(std::vector<std::pair<std::string, std::string>>& prgs, const GenotypeResult& gr, const std::vector<NovelVariant>& variants,
    std::vector<std::string>* skipped)
{
    uint32_t applied = 0;
    std::map<uint32_t, std::vector<const NovelVariant*>> by_prg;
    for (const NovelVariant& v : variants) by_prg[v.prg].push_back(&v);
    for (auto& kv : by_prg) {
        if (kv.first >= prgs.size()) continue;
        const LocusConsensus* lc = nullptr;
        for (const LocusConsensus& c : gr.consensus)
            if (c.prg == kv.first) lc = &c;
        if (!lc) continue;
        std::string& prg = prgs[kv.first].second;
        int marker = 3; // largest marker in use
        for (size_t i = 0; i < prg.size();) {
            if (std::isdigit((unsigned char)prg[i])) {
                int v = 0;
                while (i < prg.size() && std::isdigit((unsigned char)prg[i])) v = v * 10 + (prg[i++] - '0');
                marker = std::max(marker, v);
            } else ++i;
        }
        int next_marker = marker + 1 + ((marker + 1) % 2 == 0 ? 1 : 0); // the next odd number
        // where every consensus position sits in the PRG string: walk the nodes
        std::vector<const NovelVariant*> vs = kv.second;
        std::sort(vs.begin(), vs.end(), [](const NovelVariant* a, const NovelVariant* b) { return a->pos > b->pos; }); // right to left
        for (const NovelVariant* v : vs) {
            uint32_t at = 0;
            bool done = false;
            for (const ConsensusNode& n : lc->nodes) {
                const uint32_t len = n.end - n.start;
                // strictly inside the node: one base of the node stays on either side, so no marker ends up next to another
                if (len && v->pos > at && v->pos + v->ref.size() < at + len) {
                    const size_t a = n.start + (v->pos - at), b = a + v->ref.size();
                    const std::string m = std::to_string(next_marker), sep = std::to_string(next_marker + 1);
                    prg = prg.substr(0, a) + " " + m + " " + v->ref + " " + sep + " " + v->alt + " " + m + " " + prg.substr(b);
                    next_marker += 2;
                    ++applied;
                    done = true;
                    break;
                }
                at += len;
            }
            if (!done && skipped) skipped->push_back(v->chrom + ":" + std::to_string(v->pos + 1));
        }
    }
    return applied;
}

} // namespace drprg
