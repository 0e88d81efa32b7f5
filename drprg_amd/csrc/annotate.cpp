// annotate.cpp -- filter -> minor allele -> consequence -> panel / expert-rule match (predict_from_pandora_vcf).
// See report.h for the reference citations of every piece.
#include "report.h"
#include "fastx.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>

namespace drprg {
namespace report {

// ---- Prediction ---------------------------------------------------------------------------------
const char* prediction_str(Prediction p)
{
    switch (p) {
    case Prediction::None: return ".";
    case Prediction::Susceptible: return "S";
    case Prediction::Failed: return "F";
    case Prediction::MinorUnknown: return "u";
    case Prediction::Unknown: return "U";
    case Prediction::MinorResistant: return "r";
    case Prediction::Resistant: return "R";
    }
    return ".";
}
Prediction prediction_from(const std::string& s)
{
    if (s == "S") return Prediction::Susceptible;
    if (s == "F") return Prediction::Failed;
    if (s == "u") return Prediction::MinorUnknown;
    if (s == "U") return Prediction::Unknown;
    if (s == "r") return Prediction::MinorResistant;
    if (s == "R") return Prediction::Resistant;
    if (s == ".") return Prediction::None;
    throw Error(DRPRG_EFORMAT, "unknown prediction '" + s + "'");
}

// ---- Variant (/root/reference/src/panel.rs:148-287) ---------------------------------------------
Variant Variant::simplify() const
{
    if (reference == alt) return *this;
    std::string r = reference, n = alt;
    int64_t p = pos;
    while (!r.empty() && !n.empty() && r.front() == n.front() && r.size() != 1 && n.size() != 1) {
        r.erase(r.begin());
        n.erase(n.begin());
        ++p;
    }
    while (!r.empty() && !n.empty() && r.back() == n.back() && r.size() != 1 && n.size() != 1) {
        r.pop_back();
        n.pop_back();
    }
    return Variant { r, p, n };
}

std::string Variant::str() const
{
    if (reference.empty() && pos == 0 && alt == "-") return "gene_absent";
    if (reference.empty() && pos == 1 && alt == "-") return "start_lost";
    if (reference == "*" && pos >= 1 && alt == "-") return "stop_lost";
    return reference + std::to_string(pos) + alt;
}

bool Variant::parse(const std::string& s, Variant& out)
{
    // ^([a-zA-Z\*]+)(-?\d+)([a-zA-Z\*]+)$
    auto is_res = [](char c) { return std::isalpha((unsigned char)c) || c == '*'; };
    size_t i = 0;
    while (i < s.size() && is_res(s[i])) ++i;
    if (i == 0) return false;
    size_t j = i;
    if (j < s.size() && s[j] == '-') ++j;
    size_t d0 = j;
    while (j < s.size() && std::isdigit((unsigned char)s[j])) ++j;
    if (j == d0) return false;
    size_t k = j;
    while (k < s.size() && is_res(s[k])) ++k;
    if (k == j || k != s.size()) return false;
    out.reference = s.substr(0, i);
    out.pos = std::atoll(s.substr(i, j - i).c_str());
    out.alt = s.substr(j);
    return true;
}

void Variant::range(int64_t& start, int64_t& end_inclusive) const
{
    int64_t len = (int64_t)reference.size();
    int64_t e = pos + (len - 1);
    if (pos < 0 && e > -1) e += 1;
    start = pos;
    end_inclusive = e;
}

// ---- Evidence (/root/reference/src/report.rs) ----------------------------------------------------
bool Evidence::is_frameshift() const
{
    size_t a = variant.reference.size(), b = variant.alt.size();
    size_t d = a > b ? a - b : b - a;
    return !amino && d % 3 != 0;
}

std::vector<Evidence> Evidence::atomise() const
{
    if (variant.is_snp() || variant.is_indel()) return { *this };
    std::vector<Evidence> v;
    for (size_t i = 0; i < variant.reference.size() && i < variant.alt.size(); ++i) {
        Evidence e = *this;
        e.variant = Variant { std::string(1, variant.reference[i]), variant.pos + (int64_t)i, std::string(1, variant.alt[i]) };
        v.push_back(e);
    }
    return v;
}

// ---- expert rules (/root/reference/src/expert.rs) -----------------------------------------------
bool Rule::contains(const Evidence& m) const
{
    if (gene != m.gene) return false;
    int64_t rs = has_start ? start : 1, re = has_end ? end : INT64_MAX;
    int64_t ms, me;
    m.variant.range(ms, me);
    if (ms > re || rs > me) return false; // RangeInclusive::intersect
    if (type == "frameshift") return m.is_frameshift();
    if (type == "missense") return m.is_missense();
    if (type == "nonsense") return m.is_nonsense();
    return false;
}

ExpertRules load_rules(const std::string& path)
{
    ExpertRules rules;
    std::ifstream in(path);
    if (!in) return rules;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> t;
        std::stringstream ss(line);
        std::string cell;
        while (std::getline(ss, cell, ',')) t.push_back(cell);
        if (line.back() == ',') t.push_back("");
        if (t.size() != 5) throw Error(DRPRG_EFORMAT, "expert rule needs 5 comma-separated fields: " + line);
        Rule r;
        r.type = t[0];
        std::transform(r.type.begin(), r.type.end(), r.type.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        if (r.type != "frameshift" && r.type != "nonsense" && r.type != "missense" && r.type != "absence")
            throw Error(DRPRG_EFORMAT, r.type + " is not a recognised variant type");
        r.gene = t[1];
        if (!t[2].empty()) { r.has_start = true; r.start = std::atoll(t[2].c_str()); }
        if (!t[3].empty()) { r.has_end = true; r.end = std::atoll(t[3].c_str()); }
        std::stringstream ds(t[4]);
        while (std::getline(ds, cell, ';')) r.drugs.insert(cell);
        auto& v = rules[r.gene];
        bool dup = false;
        for (const Rule& o : v)
            dup |= o.type == r.type && o.has_start == r.has_start && o.has_end == r.has_end && o.start == r.start && o.end == r.end && o.drugs == r.drugs;
        if (!dup) v.push_back(r);
    }
    return rules;
}

static std::vector<Rule> rule_matches(const ExpertRules& rules, const Evidence& m)
{
    std::vector<Rule> out;
    auto it = rules.find(m.gene);
    if (it != rules.end())
        for (const Rule& r : it->second)
            if (r.contains(m)) out.push_back(r);
    return out;
}

// ---- index config (/root/reference/src/config.rs) -----------------------------------------------
IndexConfig read_config(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw Error(DRPRG_ENOENT, "Index is not valid due to missing file " + path);
    IndexConfig c;
    std::string line;
    while (std::getline(in, line)) {
        size_t e = line.find('=');
        if (e == std::string::npos) continue;
        auto trim = [](std::string s) {
            size_t a = s.find_first_not_of(" \t\"\r"), b = s.find_last_not_of(" \t\"\r");
            return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
        };
        std::string key = trim(line.substr(0, e)), val = trim(line.substr(e + 1));
        if (key == "min_match_len") c.min_match_len = std::atoi(val.c_str());
        else if (key == "max_nesting") c.max_nesting = std::atoi(val.c_str());
        else if (key == "k") c.k = std::atoi(val.c_str());
        else if (key == "w") c.w = std::atoi(val.c_str());
        else if (key == "padding") c.padding = std::atoi(val.c_str());
        else if (key == "version") c.version = val;
    }
    return c;
}

// ---- consequence (/root/reference/src/consequence.rs:79-197) ------------------------------------
static const char* codon_aa(const std::string& c)
{
    static const char* bases = "TCAG";
    static const char* table = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    int idx = 0;
    for (int i = 0; i < 3; ++i) {
        const char* p = std::strchr(bases, c[(size_t)i]);
        if (!p || !c[(size_t)i]) return nullptr;
        idx = idx * 4 + (int)(p - bases);
    }
    static char buf[64][2];
    buf[idx][0] = table[idx];
    buf[idx][1] = 0;
    return buf[idx];
}

Evidence consequence_of_variant(const VcfRecord& rec, int64_t padding, const std::string& gene_name, const std::string& gene_seq)
{
    if (rec.chrom != gene_name) throw Error(DRPRG_EINVAL, "Contig names don't match");
    std::string ref_allele = rec.alleles[0];
    size_t alt_idx = (size_t)std::max(rec.called_allele(), 0);
    if (alt_idx >= rec.alleles.size()) throw Error(DRPRG_EFORMAT, "genotype index beyond the alleles of " + rec.chrom);
    std::string alt_allele = rec.alleles[alt_idx];
    const bool is_indel = ref_allele.size() != alt_allele.size();
    if (rec.pos < 0 || (size_t)(rec.pos + rec.rlen()) > gene_seq.size())
        throw Error(DRPRG_EFORMAT, "Could not get gene reference sequence");
    std::string at_pos = gene_seq.substr((size_t)rec.pos, (size_t)rec.rlen());
    if (at_pos != ref_allele)
        throw Error(DRPRG_EFORMAT, "Reference allele " + ref_allele + " at position " + std::to_string(rec.pos + 1) + " doesn't match gene ("
                + gene_name + ") sequence " + at_pos);
    // vcf pos is 0-based; norm_pos is 1-based with no zero (…, -2, -1, 1, 2, …)
    int64_t norm_pos = rec.pos < padding ? rec.pos - padding : rec.pos - (padding - 1);
    const int64_t gene_len = (int64_t)gene_seq.size() - padding * 2;
    const bool crosses_end = (norm_pos - 1) + (int64_t)ref_allele.size() > gene_len;
    Variant variant = Variant { ref_allele, norm_pos, alt_allele }.simplify();
    Evidence ev;
    ev.gene = gene_name;
    ev.vcfid = rec.id;
    if (variant.pos < 0 || crosses_end || is_indel) {
        ev.variant = variant;
        ev.amino = false;
        return ev;
    }
    ref_allele = variant.reference;
    alt_allele = variant.alt;
    const bool adjust = norm_pos < 0 && !(variant.pos < 0);
    norm_pos = variant.pos;
    if (adjust) norm_pos += 1;
    const std::string cds = gene_seq.substr((size_t)padding, (size_t)gene_len);
    const int64_t codon_start = (norm_pos - 1) / 3 * 3;
    const int64_t codon_end = ((norm_pos - 1) + (int64_t)ref_allele.size() - 1) / 3 * 3 + 3;
    if (codon_start < 0 || (size_t)codon_end > cds.size()) throw Error(DRPRG_EFORMAT, "Couldn't extract codon sequence from gene");
    const std::string codon_seq = cds.substr((size_t)codon_start, (size_t)(codon_end - codon_start));
    std::string mutated = codon_seq;
    mutated.replace((size_t)((norm_pos - 1) - codon_start), ref_allele.size(), alt_allele);
    std::string ref_prot, alt_prot;
    for (size_t i = 0; i + 3 <= codon_seq.size() && i + 3 <= mutated.size(); i += 3) {
        const char* r = codon_aa(codon_seq.substr(i, 3));
        const char* a = codon_aa(mutated.substr(i, 3));
        if (!r || !a) throw Error(DRPRG_EFORMAT, "codon with a non-ACGT base in " + gene_name);
        ref_prot += r;
        alt_prot += a;
    }
    ev.variant = Variant { ref_prot, (norm_pos - 1) / 3 + 1, alt_prot }.simplify();
    ev.amino = true;
    return ev;
}

// ---- filters (/root/reference/src/filter.rs) -----------------------------------------------------
static int covg_for_gt(const VcfRecord& r)
{
    std::vector<int> fc, rc;
    if (!r.coverage(fc, rc)) { fc = { 0 }; rc = { 0 }; }
    int gt = r.called_allele();
    if (gt < 0) {
        int s = 0;
        for (int x : fc) s += x;
        for (int x : rc) s += x;
        return s;
    }
    return ((size_t)gt < fc.size() ? fc[(size_t)gt] : 0) + ((size_t)gt < rc.size() ? rc[(size_t)gt] : 0);
}

static void apply_filters(const FilterOpts& f, VcfRecord& r)
{
    std::vector<std::string> tags;
    const int covg = covg_for_gt(r);
    if (covg < f.min_covg) tags.push_back("ld");
    if (covg > f.max_covg) tags.push_back("hd");
    float gc = 0.0f;
    if (!r.gt_conf(gc)) gc = 0.0f;
    if (gc < f.min_gt_conf && !approx_eq_f32(gc, f.min_gt_conf)) tags.push_back("lgc");
    { // strand bias
        std::vector<int> fc, rc;
        if (r.coverage(fc, rc)) {
            int gt = r.called_allele();
            bool have = false;
            float ratio = 0;
            if (gt == -1) {
                float tf = 0, tr = 0;
                for (int x : fc) tf += (float)x;
                for (int x : rc) tr += (float)x;
                float total = tf + tr;
                if (!approx_eq_f32(total, 0.0f)) { ratio = std::min(tf, tr) / total; have = true; }
            } else if ((size_t)gt < fc.size() && (size_t)gt < rc.size()) {
                float sum = (float)fc[(size_t)gt] + (float)rc[(size_t)gt];
                if (!approx_eq_f32(sum, 0.0f)) { ratio = std::min((float)fc[(size_t)gt], (float)rc[(size_t)gt]) / sum; have = true; }
            }
            if (have && ratio < f.min_strand_bias && !approx_eq_f32(ratio, f.min_strand_bias)) tags.push_back("sb");
        }
    }
    { // long indel
        int gt = r.called_allele();
        if (gt >= 1 && f.has_max_indel) {
            int64_t l = (size_t)gt < r.alleles.size() ? (int64_t)r.alleles[(size_t)gt].size() : 0;
            if (std::llabs(r.rlen() - l) > (int64_t)f.max_indel) tags.push_back("lindel");
        }
    }
    float frs;
    if (r.fraction_read_support(frs) && frs < f.min_frs && !approx_eq_f32(frs, f.min_frs)) tags.push_back("frs");
    if (tags.empty()) tags.push_back("PASS");
    r.filters = tags;
}

static std::string fmt_g(double v)
{
    char b[64];
    std::snprintf(b, sizeof b, "%g", v);
    return b;
}
static std::string fmt_fixed(double v, int prec)
{
    char b[64];
    std::snprintf(b, sizeof b, "%.*f", prec, v);
    return b;
}
// Rust's `{}` for f32: shortest representation that round-trips
static std::string fmt_f32_display(float v)
{
    for (int prec = 1; prec < 12; ++prec) {
        char b[64];
        std::snprintf(b, sizeof b, "%.*g", prec, (double)v);
        if (std::strtof(b, nullptr) == v) return b;
    }
    return fmt_g(v);
}

static void add_annotation_headers(const AnnotateOpts& o, std::vector<std::string>& h)
{
    const FilterOpts& f = o.filter;
    if (f.min_covg > -1) h.push_back("##FILTER=<ID=ld,Description=\"Kmer coverage on called allele less than " + std::to_string(f.min_covg) + "\">");
    if (f.max_covg < INT32_MAX) h.push_back("##FILTER=<ID=hd,Description=\"Kmer coverage on called allele more than " + std::to_string(f.min_covg) + "\">");
    if (f.min_strand_bias > -1.0f)
        h.push_back("##FILTER=<ID=sb,Description=\"A strand on the called allele has less than " + fmt_fixed(f.min_strand_bias * 100.0f, 2) + "% of the coverage for that allele\">");
    if (f.min_gt_conf > -1.0f) h.push_back("##FILTER=<ID=lgc,Description=\"Genotype confidence score less than " + fmt_fixed(f.min_gt_conf, 1) + "\">");
    if (f.has_max_indel) h.push_back("##FILTER=<ID=lindel,Description=\"Indel is longer than " + std::to_string(f.max_indel) + "bp\">");
    if (f.min_frs > -1.0f) h.push_back("##FILTER=<ID=frs,Description=\"Fraction of read support on called allele is less than " + fmt_f32_display(f.min_frs) + "\">");
    h.push_back("##INFO=<ID=VARID,Number=.,Type=String,Description=\"The identifier for the panel variant(s) the record overlaps with\">");
    h.push_back("##INFO=<ID=PREDICT,Number=.,Type=String,Description=\"The drug resistance prediction(s) for the corresponding VARID(s), where 'R' = resistant, 'S' = susceptible, 'F' = failed, and 'U' = unknown\">");
    h.push_back("##INFO=<ID=OGT,Number=1,Type=String,Description=\"Original genotype after adjusting for minor allele depth proportions of " + fmt_f32_display(o.minor.maf) + "\">");
    h.push_back("##INFO=<ID=PDP,Number=R,Type=Float,Description=\"Proportion of the total position depth found on this allele\">");
}

// ---- minor allele (/root/reference/src/minor.rs:70-149) ------------------------------------------
static bool parse_floats(const std::string* s, std::vector<float>& out)
{
    if (!s) return false;
    out.clear();
    std::stringstream ss(*s);
    std::string t;
    while (std::getline(ss, t, ',')) out.push_back(t == "." ? 0.0f : std::strtof(t.c_str(), nullptr));
    return true;
}

static int check_for_minor_alternate(const MinorOpts& m, VcfRecord& r)
{
    std::vector<float> props;
    const bool have = r.depth_proportions(props);
    if (have) {
        std::string s;
        for (size_t i = 0; i < props.size(); ++i) s += (i ? "," : "") + fmt_g(props[i]);
        r.set_info("PDP", s);
    }
    const int gt = r.called_allele();
    if (r.alleles.size() < 2 || !have || gt < 0) return -1;
    std::vector<float> gaps;
    if (!parse_floats(r.fmt("GAPS"), gaps) || (size_t)gt >= gaps.size())
        throw Error(DRPRG_EFORMAT, "Failed to check for minor allele in record " + r.chrom + ":" + std::to_string(r.pos));
    std::vector<size_t> ix(props.size());
    for (size_t i = 0; i < ix.size(); ++i) ix[i] = i;
    std::stable_sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return props[a] < props[b]; });
    if (gaps[(size_t)gt] > m.max_called_gaps) return -1;
    int found = -1;
    for (size_t q = ix.size(); q-- > 0;) {
        size_t i = ix[q];
        if ((int)i == gt) continue;
        float g = i < gaps.size() ? gaps[i] : 0.0f;
        float gd = g - gaps[(size_t)gt];
        if (props[i] >= m.maf && g <= m.max_gaps && gd <= m.max_gaps_diff) {
            found = (int)i;
            break;
        }
    }
    if (found < 0) return -1;
    std::vector<int> fc, rc;
    if (!r.coverage(fc, rc)) { fc = { 0 }; rc = { 0 }; }
    const size_t g = (size_t)found;
    const float sum = (float)(fc[g] + rc[g]);
    const bool low = (fc[g] + rc[g]) < m.minor_min_covg;
    const bool sb = approx_eq_f32(sum, 0.0f) ? true : (float)std::min(fc[g], rc[g]) / sum < m.minor_min_strand_bias;
    return (low || sb) ? -1 : found;
}

// ---- panel / rules matching (/root/reference/src/predict.rs:546-679) ------------------------------
struct Annotator {
    const AnnotateOpts& o;
    std::vector<PanelRecordBcf> panel;
    std::map<std::string, const PanelRecordBcf*> by_id;
    ExpertRules rules;
    std::map<std::string, std::string> genes;
    int64_t padding;

    std::vector<const PanelRecordBcf*> fetch(const std::string& chrom, int64_t start, int64_t end) const
    {
        std::vector<const PanelRecordBcf*> out; // records overlapping [start, end), file order
        for (const PanelRecordBcf& p : panel) {
            if (p.rec.chrom != chrom) continue;
            int64_t pe = p.rec.pos + std::max<int64_t>(p.rec.rlen(), 1);
            if (p.rec.pos < end && pe > start) out.push_back(&p);
        }
        return out;
    }

    Evidence consequence(const VcfRecord& r) const
    {
        auto it = genes.find(r.chrom);
        if (it == genes.end()) throw Error(DRPRG_EFORMAT, "Couldn't find gene " + r.chrom + " in index FASTA");
        return consequence_of_variant(r, padding, r.chrom, it->second);
    }

    void against_index(const VcfRecord& r, const std::vector<const PanelRecordBcf*>& hits, const std::vector<Evidence>& csqs,
        std::vector<std::string>& muts, std::vector<Prediction>& preds) const
    {
        for (const PanelRecordBcf* p : hits) {
            const std::string& vid = p->rec.id;
            size_t us = vid.find('_');
            if (us == std::string::npos) throw Error(DRPRG_EFORMAT, "Couldn't split variant ID " + vid + " at underscore");
            Variant vv;
            if (!Variant::parse(vid.substr(us + 1), vv)) throw Error(DRPRG_EFORMAT, "bad panel variant " + vid);
            const bool none_drug = std::find(p->drugs.begin(), p->drugs.end(), "NONE") != p->drugs.end();
            Prediction pred = Prediction::None;
            if (r.called_allele() == -1) {
                pred = Prediction::Failed;
            } else {
                for (const Evidence& c : csqs) {
                    if (c.variant.pos != vv.pos) continue;
                    const bool is_x = !vid.empty() && vid.back() == 'X';
                    bool match;
                    if (is_x) {
                        match = c.amino ? (c.variant.reference != c.variant.alt && c.variant.alt != "*") : (c.variant.reference != c.variant.alt);
                    } else {
                        match = c.variant_string() == vid;
                    }
                    if (match) {
                        pred = none_drug ? Prediction::Susceptible : Prediction::Resistant;
                        break;
                    }
                }
                if (pred < Prediction::Resistant) {
                    int m = r.argmatch(p->rec);
                    if (m > 0) pred = none_drug ? Prediction::Susceptible : Prediction::Resistant;
                }
            }
            preds.push_back(pred);
            muts.push_back(vid);
        }
    }

    void against_rules(const VcfRecord& r, const std::vector<Evidence>& csqs, std::vector<std::string>& muts,
        std::vector<Prediction>& preds) const
    {
        for (const Evidence& c : csqs) {
            Prediction pred = Prediction::Susceptible;
            std::vector<Rule> ms = rule_matches(rules, c);
            if (ms.empty()) continue;
            for (const Rule& rule : ms) {
                if (!rule.drugs.count("NONE")) {
                    int ca = r.called_allele();
                    pred = ca == -1 ? Prediction::Failed : (ca > 0 ? Prediction::Resistant : Prediction::None);
                    break;
                }
            }
            muts.push_back(c.variant_string());
            preds.push_back(pred);
        }
    }

    void record_predictions(const VcfRecord& r, const std::vector<const PanelRecordBcf*>& hits, const std::vector<Evidence>& csqs,
        std::vector<std::string>& muts, std::vector<Prediction>& preds) const
    {
        against_index(r, hits, csqs, muts, preds);
        against_rules(r, csqs, muts, preds);
        Prediction mx = Prediction::None;
        for (Prediction p : preds) mx = std::max(mx, p);
        if (mx == Prediction::None && r.called_allele() > 0) {
            for (const Evidence& c : csqs) {
                muts.push_back(c.variant_string());
                preds.push_back(c.is_synonymous() && o.ignore_synonymous ? Prediction::None : Prediction::Unknown);
            }
        }
    }
};

static void dedup(std::vector<std::string>& muts, std::vector<Prediction>& preds)
{
    // the reference goes through a HashMap (arbitrary order); first-appearance order is kept here
    std::vector<std::string> m2;
    std::vector<Prediction> p2;
    for (size_t i = 0; i < muts.size(); ++i) {
        auto it = std::find(m2.begin(), m2.end(), muts[i]);
        if (it == m2.end()) {
            m2.push_back(muts[i]);
            p2.push_back(preds[i]);
        } else {
            size_t j = (size_t)(it - m2.begin());
            p2[j] = std::max(p2[j], preds[i]);
        }
    }
    muts.swap(m2);
    preds.swap(p2);
}

void annotate_vcf(const IndexFiles& idx, const std::string& pandora_vcf, const std::string& out_vcf, const AnnotateOpts& o)
{
    VcfFile in = read_vcf(pandora_vcf);
    Annotator an { o, read_panel_bcf(idx.panel_bcf()), {}, load_rules(idx.rules_csv()), {}, 0 };
    for (auto& kv : read_fasta(idx.genes_fa())) an.genes[kv.first] = kv.second;
    an.padding = read_config(idx.config()).padding;
    std::mt19937_64 rng(o.id_seed ? o.id_seed : std::random_device {}());
    std::set<std::string> used_ids;

    std::ofstream out(out_vcf);
    if (!out) throw Error(DRPRG_EIO, "cannot write " + out_vcf);
    std::vector<std::string> header = in.header;
    add_annotation_headers(o, header);
    for (const std::string& h : header) out << h << "\n";
    out << in.column_line << "\n";

    for (VcfRecord& r : in.records) {
        float gc;
        if (r.has_no_depth() && r.gt_conf(gc) && gc == 0.0f) r.set_fmt("GT", ".");
        apply_filters(o.filter, r);
        std::string id;
        do {
            char b[16];
            std::snprintf(b, sizeof b, "%08x", (unsigned)(rng() & 0xFFFFFFFFu));
            id = b;
        } while (!used_ids.insert(id).second);
        r.id = id;
        bool known_contig = false;
        for (const PanelRecordBcf& p : an.panel) known_contig |= p.rec.chrom == r.chrom;
        if (!known_contig) { // unwrap_or_continue!(name2rid): the record is dropped
            continue;
        }
        std::vector<const PanelRecordBcf*> hits = an.fetch(r.chrom, r.pos, r.end());
        std::vector<Evidence> csqs = an.consequence(r).atomise();
        std::vector<std::string> muts;
        std::vector<Prediction> preds;
        an.record_predictions(r, hits, csqs, muts, preds);
        Prediction max_pred = Prediction::None;
        for (Prediction p : preds) max_pred = std::max(max_pred, p);

        int minor = check_for_minor_alternate(o.minor, r);
        if (minor > 0 && max_pred < Prediction::Resistant) {
            const std::string ogt = std::to_string(r.called_allele());
            r.set_info("OGT", ogt);
            r.set_fmt("GT", std::to_string(minor));
            std::vector<Evidence> csqs2 = an.consequence(r).atomise();
            std::vector<std::string> muts2;
            std::vector<Prediction> preds2;
            an.record_predictions(r, hits, csqs2, muts2, preds2);
            for (Prediction& p : preds2) {
                if (p == Prediction::Unknown) p = Prediction::MinorUnknown;
                else if (p == Prediction::Resistant) p = Prediction::MinorResistant;
            }
            Prediction max_minor = Prediction::None;
            for (Prediction p : preds2) max_minor = std::max(max_minor, p);
            if (max_minor < max_pred) { // undo the genotype adjustment
                r.set_fmt("GT", ogt);
                r.clear_info("OGT");
            }
            muts.insert(muts.end(), muts2.begin(), muts2.end());
            preds.insert(preds.end(), preds2.begin(), preds2.end());
        }
        dedup(muts, preds);
        r.clear_info("VARID");
        r.clear_info("PREDICT");
        if (!muts.empty()) {
            std::string ms, ps;
            for (size_t i = 0; i < muts.size(); ++i) {
                ms += (i ? "," : "") + muts[i];
                ps += (i ? "," : "") + std::string(prediction_str(preds[i]));
            }
            r.set_info("VARID", ms);
            r.set_info("PREDICT", ps);
        }
        out << r.to_line() << "\n";
    }
    if (!out) throw Error(DRPRG_EIO, "short write to " + out_vcf);
}

} // namespace report
} // namespace drprg
