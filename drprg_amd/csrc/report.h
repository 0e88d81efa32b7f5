// report.h -- post-VCF stage of `drprg predict`: filter -> minor allele -> consequence -> panel / expert-rule
// match -> annotated VCF -> susceptibility JSON (NEXT-1).
//
// C++ re-expression of /root/reference/src/predict.rs:420-1139 (predict_from_pandora_vcf, vcf_to_json),
// src/lib.rs:935-1181 (VcfExt), src/filter.rs, src/minor.rs, src/consequence.rs, src/expert.rs,
// src/panel.rs:148-287 (Variant, Residue) and src/report.rs.  Pinned by the reference's golden files
// (tests/golden/downstream/: in*.vcf -> out*.vcf -> expected*.json).
#pragma once
#include "common.h"
#include <map>
#include <set>

namespace drprg {
namespace report {

// ---- VCF text model (single sample) -------------------------------------------------------------
struct VcfRecord {
    std::string chrom;
    int64_t pos = 0; // 0-based
    std::string id = ".";
    std::vector<std::string> alleles; // REF first
    std::string qual = ".";
    std::vector<std::string> filters; // empty = "."
    std::vector<std::pair<std::string, std::string>> info; // ordered; flag = empty value
    std::vector<std::string> format;
    std::vector<std::string> sample; // one value per FORMAT key

    int64_t rlen() const { return (int64_t)alleles[0].size(); }
    int64_t end() const { return pos + rlen(); }
    const std::string* fmt(const std::string& key) const;
    void set_fmt(const std::string& key, const std::string& value);
    const std::string* get_info(const std::string& key) const;
    void set_info(const std::string& key, const std::string& value);
    void clear_info(const std::string& key);
    // VcfExt (/root/reference/src/lib.rs:935-1181)
    int called_allele() const;
    bool coverage(std::vector<int>& fwd, std::vector<int>& rev) const;
    bool gt_conf(float& out) const;
    bool fraction_read_support(float& out) const;
    bool depth_proportions(std::vector<float>& out) const;
    bool has_no_depth() const;
    bool is_pass() const;
    bool is_indel() const;
    std::string slice(int64_t start, int64_t stop, int allele /* -1 = called (REF when null) */) const;
    int argmatch(const VcfRecord& other) const; // -1 = None
    std::string to_line() const;
};

struct VcfFile {
    std::vector<std::string> header; // every ## line
    std::string column_line;         // #CHROM ...
    std::vector<VcfRecord> records;
    std::vector<std::string> contigs() const;
};
VcfFile read_vcf(const std::string& path); // plain or gz text VCF
// BCF2.2 in BGZF, what the reference writes through rust-htslib (/root/reference/src/predict.rs:429-431); bcfout.cpp
void write_bcf(const std::string& path, const VcfFile& vcf);
void vcf_to_bcf(const std::string& vcf_path, const std::string& bcf_path);
VcfRecord parse_vcf_line(const std::string& line);

// ---- panel.bcf (BCF2 in BGZF) -------------------------------------------------------------------
struct PanelRecordBcf {
    VcfRecord rec; // chrom, pos, id, alleles
    std::vector<std::string> drugs;
    std::string residue; // DNA | PROT
};
std::vector<PanelRecordBcf> read_panel_bcf(const std::string& path);

// ---- domain types -------------------------------------------------------------------------------
enum class Prediction { None = 0, Susceptible, Failed, MinorUnknown, Unknown, MinorResistant, Resistant };
const char* prediction_str(Prediction p);
Prediction prediction_from(const std::string& s);

struct Variant {
    std::string reference;
    int64_t pos = 0;
    std::string alt;
    Variant simplify() const;
    bool is_indel() const { return reference.size() != alt.size(); }
    bool is_snp() const { return reference.size() == 1 && alt.size() == 1; }
    std::string str() const;
    static bool parse(const std::string& s, Variant& out);
    void range(int64_t& start, int64_t& end_inclusive) const;
};

struct Evidence {
    Variant variant;
    std::string gene;
    bool amino = false; // Residue::Amino ("PROT") vs Nucleic ("DNA")
    std::string vcfid;
    std::string variant_string() const { return gene + "_" + variant.str(); }
    bool is_synonymous() const { return amino && variant.reference == variant.alt; }
    bool is_nonsense() const { return amino && variant.alt == "*"; }
    bool is_missense() const { return amino && !is_nonsense() && !is_synonymous(); }
    bool is_frameshift() const;
    std::vector<Evidence> atomise() const;
};

struct Rule {
    std::string type; // frameshift | nonsense | missense | absence
    std::string gene;
    bool has_start = false, has_end = false;
    int64_t start = 0, end = 0;
    std::set<std::string> drugs;
    bool contains(const Evidence& e) const;
};
using ExpertRules = std::map<std::string, std::vector<Rule>>;
ExpertRules load_rules(const std::string& csv_path); // missing file -> empty

struct FilterOpts { // /root/reference/src/filter.rs:165-197; Default = everything disabled
    int min_covg = -1;
    int max_covg = INT32_MAX;
    float min_strand_bias = -1.0f;
    float min_gt_conf = -1.0f;
    bool has_max_indel = false;
    int max_indel = 0;
    float min_frs = -1.0f;
};
struct MinorOpts { // /root/reference/src/minor.rs:19-49; Default derive = all zero
    float maf = 0, max_gaps = 0, max_called_gaps = 0, max_gaps_diff = 0;
    int minor_min_covg = 0;
    float minor_min_strand_bias = 0;
};
struct AnnotateOpts {
    FilterOpts filter;
    MinorOpts minor;
    bool ignore_synonymous = false;
    uint64_t id_seed = 0; // 0 = random IDs (the reference uses Uuid::new_v4()[..8])
};

struct IndexFiles { // the files validate_index requires (/root/reference/src/predict.rs:400-418)
    std::string dir;
    std::string config() const { return dir + "/.config.toml"; }
    std::string panel_bcf() const { return dir + "/panel.bcf"; }
    std::string genes_fa() const { return dir + "/genes.fa"; }
    std::string rules_csv() const { return dir + "/rules.csv"; }
    std::string prg() const { return dir + "/dr.prg"; }
};
struct IndexConfig {
    int min_match_len = 5, max_nesting = 5, k = 15, w = 11, padding = 100;
    std::string version = "unknown";
};
IndexConfig read_config(const std::string& toml_path);

Evidence consequence_of_variant(const VcfRecord& rec, int64_t padding, const std::string& gene_name, const std::string& gene_seq);

// predict_from_pandora_vcf (/root/reference/src/predict.rs:420-544): writes the annotated VCF (text)
void annotate_vcf(const IndexFiles& idx, const std::string& pandora_vcf, const std::string& out_vcf, const AnnotateOpts& o);
// vcf_to_json (/root/reference/src/predict.rs:716-1086)
void vcf_to_json(const IndexFiles& idx, const std::string& vcf_path, const std::string& json_path, const std::string& sample,
    int padding, const std::string& index_version);

bool approx_eq_f32(float a, float b); // float_cmp::approx_eq!(f32, ..) defaults: epsilon, 4 ulps

} // namespace report
} // namespace drprg
