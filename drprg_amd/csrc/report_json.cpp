// report_json.cpp -- vcf_to_json: annotated VCF -> susceptibility JSON (/root/reference/src/predict.rs:716-1086).
// Output layout = serde_json::to_string_pretty of BTreeMaps: keys sorted, two-space indent.
#include "report.h"
#include "fastx.h"
#include <algorithm>
#include <fstream>
#include <sstream>

namespace drprg {
namespace report {

namespace {

struct Susceptibility {
    Prediction predict = Prediction::Susceptible;
    std::vector<Evidence> evidence;
};

std::string jstr(const std::string& s)
{
    std::string o = "\"";
    for (char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\t': o += "\\t"; break;
        case '\r': o += "\\r"; break;
        default:
            if ((unsigned char)c < 0x20) {
                char b[8];
                std::snprintf(b, sizeof b, "\\u%04x", c);
                o += b;
            } else o += c;
        }
    }
    return o + "\"";
}

std::string jlist(const std::vector<std::string>& v, int indent)
{
    if (v.empty()) return "[]";
    std::string pad(indent + 2, ' '), o = "[\n";
    for (size_t i = 0; i < v.size(); ++i) o += pad + jstr(v[i]) + (i + 1 < v.size() ? ",\n" : "\n");
    return o + std::string(indent, ' ') + "]";
}

std::vector<std::string> split_commas(const std::string& s)
{
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string t;
    while (std::getline(ss, t, ',')) out.push_back(t);
    return out;
}

void set_resistant(Susceptibility& e, const Evidence& ev)
{
    if (e.predict == Prediction::Resistant) e.evidence.push_back(ev);
    else {
        e.predict = Prediction::Resistant;
        e.evidence = { ev };
    }
}

} // namespace

void vcf_to_json(const IndexFiles& idx, const std::string& vcf_path, const std::string& json_path, const std::string& sample,
    int padding, const std::string& index_version)
{
    // panel: variant id -> (drugs, residue)
    std::map<std::string, std::pair<std::set<std::string>, bool>> var2drugs;
    for (const PanelRecordBcf& p : read_panel_bcf(idx.panel_bcf())) {
        if (p.drugs.empty()) continue;
        var2drugs[p.rec.id] = { std::set<std::string>(p.drugs.begin(), p.drugs.end()), p.residue == "PROT" };
    }
    std::map<std::string, std::set<std::string>> gene2drugs;
    for (auto& kv : var2drugs) {
        size_t us = kv.first.find('_');
        if (us == std::string::npos) throw Error(DRPRG_EFORMAT, "Couldn't split variant ID " + kv.first + " at underscore");
        auto& e = gene2drugs[kv.first.substr(0, us)];
        e.insert(kv.second.first.begin(), kv.second.first.end());
    }
    const ExpertRules rules = load_rules(idx.rules_csv());
    for (auto& gr : rules)
        for (const Rule& r : gr.second) gene2drugs[gr.first].insert(r.drugs.begin(), r.drugs.end());

    std::map<std::string, std::string> genes;
    for (auto& kv : read_fasta(idx.genes_fa())) genes[kv.first] = kv.second;
    auto consequence = [&](const VcfRecord& r) {
        auto it = genes.find(r.chrom);
        if (it == genes.end()) throw Error(DRPRG_EFORMAT, "Couldn't find gene " + r.chrom + " in index FASTA");
        return consequence_of_variant(r, padding, r.chrom, it->second);
    };

    std::map<std::string, Susceptibility> json;
    VcfFile vcf = read_vcf(vcf_path);
    std::set<std::string> present;
    for (const std::string& c : vcf.contigs()) present.insert(c);
    std::set<std::string> absent;
    for (auto& kv : gene2drugs)
        if (!present.count(kv.first)) absent.insert(kv.first);

    // absent genes with an "absence" expert rule
    for (auto& gr : rules) {
        if (!absent.count(gr.first)) continue;
        for (const Rule& rule : gr.second) {
            if (rule.type != "absence") continue;
            for (const std::string& drug : rule.drugs) {
                if (drug == "NONE") continue;
                Evidence ev;
                ev.variant = Variant { "", 0, "-" };
                ev.gene = gr.first;
                set_resistant(json[drug], ev);
            }
        }
    }
    // present genes whose start loss counts as absence
    std::map<std::string, std::vector<std::string>> check_for_start_loss;
    for (const std::string& gene : present) {
        auto it = rules.find(gene);
        if (it == rules.end()) continue;
        for (const Rule& r : it->second)
            if (r.type == "absence") {
                check_for_start_loss[gene] = std::vector<std::string>(r.drugs.begin(), r.drugs.end());
                break;
            }
    }
    struct NullIv {
        bool some;
        int64_t start, end;
        std::string id;
    };
    std::map<std::string, std::vector<NullIv>> null_intervals;

    for (size_t i = 0; i < vcf.records.size(); ++i) {
        const VcfRecord& record = vcf.records[i];
        const bool is_alt = record.called_allele() > 0;
        std::vector<Prediction> preds;
        if (const std::string* s = record.get_info("PREDICT"))
            for (auto& t : split_commas(*s)) preds.push_back(prediction_from(t));
        if (preds.empty() && is_alt) throw Error(DRPRG_EFORMAT, "PREDICT tag is unexpectedly empty in VCF");
        std::vector<std::string> varids;
        if (const std::string* s = record.get_info("VARID")) varids = split_commas(*s);
        if (varids.empty() && is_alt) throw Error(DRPRG_EFORMAT, "VARID tag is unexpectedly empty in VCF");
        Prediction max_pred = Prediction::None;
        for (Prediction p : preds) max_pred = std::max(max_pred, p);
        const bool is_failed = max_pred == Prediction::Failed || record.called_allele() < 0;
        null_intervals[record.chrom].push_back(NullIv { is_failed, record.pos, record.end(), record.id });
        if ((!record.is_pass() && !is_failed) || max_pred == Prediction::None) continue;

        for (size_t q = 0; q < preds.size() && q < varids.size(); ++q) {
            if (preds[q] != max_pred) continue;
            const std::string& varid = varids[q];
            size_t us = varid.find('_');
            if (us == std::string::npos) throw Error(DRPRG_EFORMAT, "Couldn't split variant ID " + varid + " at underscore");
            const std::string chrom = varid.substr(0, us), var = varid.substr(us + 1);
            std::set<std::string> drugs;
            bool amino = false;
            auto it = var2drugs.find(varid);
            if (it != var2drugs.end()) {
                drugs = it->second.first;
                amino = it->second.second;
            } else {
                bool have_res = false;
                for (const Evidence& csq : consequence(record).atomise()) {
                    if (csq.variant_string() != varid) continue;
                    auto rit = rules.find(csq.gene);
                    if (rit != rules.end())
                        for (const Rule& rule : rit->second)
                            if (rule.contains(csq)) drugs.insert(rule.drugs.begin(), rule.drugs.end());
                    amino = csq.amino;
                    have_res = true;
                    break;
                }
                if (drugs.empty()) {
                    auto g = gene2drugs.find(chrom);
                    if (g != gene2drugs.end()) drugs = g->second;
                }
                if (!have_res) throw Error(DRPRG_EFORMAT, "Could not find variant " + varid + " in panel or expert rules");
            }
            Evidence ev;
            if (!Variant::parse(var, ev.variant)) throw Error(DRPRG_EFORMAT, "The variant is not in the correct format [<STR><INT><STR>]: " + var);
            ev.gene = chrom;
            ev.amino = amino;
            ev.vcfid = record.id;
            for (const std::string& drug : drugs) {
                if (drug == "NONE") continue;
                Susceptibility& e = json[drug];
                if (e.predict < preds[q]) {
                    e.predict = preds[q];
                    e.evidence = { ev };
                } else if (e.predict == preds[q]) {
                    e.evidence.push_back(ev);
                }
            }
        }
    }

    std::map<std::string, int64_t> gene_lengths;
    for (auto& kv : genes) gene_lengths[kv.first] = (int64_t)kv.second.size();

    for (auto& gi : null_intervals) { // (sorted gene order; the reference iterates a HashMap)
        const std::string& gene = gi.first;
        auto gl = gene_lengths.find(gene);
        if (gl == gene_lengths.end()) throw Error(DRPRG_EFORMAT, "gene " + gene + " of the VCF is not in the index FASTA");
        const int64_t stop_pos = gl->second - (int64_t)padding;
        bool have_start = false, spans_start = false, spans_stop = false;
        int64_t current_start = 0;
        std::vector<std::string> start_ids, stop_ids;
        for (const NullIv& el : gi.second) {
            if (el.some) {
                start_ids.push_back(el.id);
                stop_ids.push_back(el.id);
                if (!have_start) {
                    current_start = el.start;
                    have_start = true;
                }
                if (current_start <= padding && padding < el.end) spans_start = true;
                if (current_start <= stop_pos && stop_pos < el.end) spans_stop = true;
            } else {
                have_start = false;
                if (!spans_start) start_ids.clear();
                if (!spans_stop) stop_ids.clear();
            }
        }
        auto join = [](const std::vector<std::string>& v) {
            std::string s;
            for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + v[i];
            return s;
        };
        if (spans_start) {
            auto it = check_for_start_loss.find(gene);
            if (it != check_for_start_loss.end())
                for (const std::string& drug : it->second) {
                    if (drug == "NONE") continue;
                    Evidence ev;
                    ev.variant = Variant { "", 1, "-" };
                    ev.gene = gene;
                    ev.vcfid = join(start_ids);
                    set_resistant(json[drug], ev);
                }
        }
        if (spans_stop) {
            auto it = gene2drugs.find(gene);
            if (it != gene2drugs.end())
                for (const std::string& drug : it->second) {
                    if (drug == "NONE") continue;
                    Evidence ev;
                    ev.variant = Variant { "*", gl->second, "-" };
                    ev.gene = gene;
                    ev.vcfid = join(stop_ids);
                    Susceptibility& e = json[drug];
                    if (e.predict == Prediction::Unknown) e.evidence.push_back(ev);
                    else if (e.predict < Prediction::Unknown) {
                        e.predict = Prediction::Unknown;
                        e.evidence = { ev };
                    }
                }
        }
    }
    for (auto& kv : var2drugs)
        for (const std::string& d : kv.second.first)
            if (d != "NONE") json[d];

    std::ostringstream o;
    o << "{\n  \"genes\": {\n";
    o << "    \"absent\": " << jlist(std::vector<std::string>(absent.begin(), absent.end()), 4) << ",\n";
    o << "    \"present\": " << jlist(std::vector<std::string>(present.begin(), present.end()), 4) << "\n  },\n";
    o << "  \"sample\": " << jstr(sample) << ",\n";
    o << "  \"susceptibility\": {";
    size_t n = 0;
    for (auto& kv : json) {
        o << (n++ ? ",\n" : "\n") << "    " << jstr(kv.first) << ": {\n      \"evidence\": ";
        if (kv.second.evidence.empty()) o << "[]";
        else {
            o << "[\n";
            for (size_t i = 0; i < kv.second.evidence.size(); ++i) {
                const Evidence& e = kv.second.evidence[i];
                o << "        {\n          \"gene\": " << jstr(e.gene) << ",\n          \"residue\": " << jstr(e.amino ? "PROT" : "DNA")
                  << ",\n          \"variant\": " << jstr(e.variant.str()) << ",\n          \"vcfid\": " << jstr(e.vcfid) << "\n        }"
                  << (i + 1 < kv.second.evidence.size() ? ",\n" : "\n");
            }
            o << "      ]";
        }
        o << ",\n      \"predict\": " << jstr(prediction_str(kv.second.predict)) << "\n    }";
    }
    o << (json.empty() ? "}" : "\n  }") << ",\n";
    o << "  \"version\": {\n    \"drprg\": \"0.1.1\",\n    \"index\": " << jstr(index_version) << "\n  }\n}";
    std::ofstream f(json_path);
    if (!f) throw Error(DRPRG_EIO, "cannot write " + json_path);
    f << o.str();
    if (!f) throw Error(DRPRG_EIO, "short write to " + json_path);
}

} // namespace report
} // namespace drprg
