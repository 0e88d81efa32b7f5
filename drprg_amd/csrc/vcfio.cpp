// vcfio.cpp -- VCF text model with the VcfExt helpers of /root/reference/src/lib.rs:935-1181, and a minimal
// BGZF + BCF2 reader for the index's panel.bcf (no htslib in this image).
#include "report.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <zlib.h>

namespace drprg {
namespace report {

static std::vector<std::string> split(const std::string& s, char d)
{
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        size_t b = s.find(d, a);
        if (b == std::string::npos) {
            out.push_back(s.substr(a));
            break;
        }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

bool approx_eq_f32(float a, float b)
{
    if (a == b) return true;
    if (std::fabs(a - b) <= 1.1920929e-7f) return true; // f32::EPSILON
    int32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if ((ia < 0) != (ib < 0)) return false;
    int64_t d = (int64_t)ia - (int64_t)ib;
    return (d < 0 ? -d : d) <= 4;
}

const std::string* VcfRecord::fmt(const std::string& key) const
{
    for (size_t i = 0; i < format.size() && i < sample.size(); ++i)
        if (format[i] == key) return &sample[i];
    return nullptr;
}
void VcfRecord::set_fmt(const std::string& key, const std::string& value)
{
    for (size_t i = 0; i < format.size(); ++i)
        if (format[i] == key) {
            if (sample.size() <= i) sample.resize(i + 1, ".");
            sample[i] = value;
            return;
        }
    format.push_back(key);
    sample.resize(format.size(), ".");
    sample.back() = value;
}
const std::string* VcfRecord::get_info(const std::string& key) const
{
    for (auto& kv : info)
        if (kv.first == key) return &kv.second;
    return nullptr;
}
void VcfRecord::set_info(const std::string& key, const std::string& value)
{
    for (auto& kv : info)
        if (kv.first == key) {
            kv.second = value;
            return;
        }
    info.emplace_back(key, value);
}
void VcfRecord::clear_info(const std::string& key)
{
    info.erase(std::remove_if(info.begin(), info.end(), [&](const auto& kv) { return kv.first == key; }), info.end());
}

int VcfRecord::called_allele() const
{
    const std::string* gt = fmt("GT");
    if (!gt || gt->empty()) return -1;
    // a single haploid allele; anything else ("." , "0/1", ...) is null as in the reference
    for (char c : *gt)
        if (c < '0' || c > '9') return -1;
    return std::atoi(gt->c_str());
}

static bool parse_ints(const std::string* s, std::vector<int>& out)
{
    if (!s) return false;
    out.clear();
    for (const std::string& t : split(*s, ',')) {
        if (t == "." || t.empty()) out.push_back(0);
        else out.push_back(std::atoi(t.c_str()));
    }
    return true;
}

bool VcfRecord::coverage(std::vector<int>& fwd, std::vector<int>& rev) const
{
    return parse_ints(fmt("MEAN_FWD_COVG"), fwd) && parse_ints(fmt("MEAN_REV_COVG"), rev);
}

bool VcfRecord::gt_conf(float& out) const
{
    const std::string* s = fmt("GT_CONF");
    if (!s || *s == ".") return false;
    out = std::strtof(s->c_str(), nullptr);
    return true;
}

bool VcfRecord::fraction_read_support(float& out) const
{
    std::vector<int> fc, rc;
    if (!coverage(fc, rc)) return false;
    if (fc.size() < 2) {
        out = 1.0f;
        return true;
    }
    int gt = called_allele();
    if (gt < 0) return false;
    float called = (float)(fc[(size_t)gt] + rc[(size_t)gt]);
    int other = 0;
    if (gt > 0) other = fc[0] + rc[0];
    else
        for (size_t i = 0; i < fc.size(); ++i)
            if ((int)i != gt) other = std::max(other, fc[i] + rc[i]);
    float f = called / (called + (float)other);
    if (std::isnan(f)) return false;
    out = f;
    return true;
}

bool VcfRecord::depth_proportions(std::vector<float>& out) const
{
    std::vector<int> fc, rc;
    if (!coverage(fc, rc)) return false;
    float total = 0;
    std::vector<float> d;
    for (size_t i = 0; i < fc.size(); ++i) {
        d.push_back((float)(fc[i] + rc[i]));
        total += d.back();
    }
    if (total == 0.0f) return false;
    out.clear();
    for (float x : d) out.push_back(x / total);
    return true;
}

bool VcfRecord::has_no_depth() const
{
    std::vector<int> fc, rc;
    if (!coverage(fc, rc)) return true;
    long total = 0;
    for (int x : fc) total += x;
    for (int x : rc) total += x;
    return total == 0;
}

bool VcfRecord::is_pass() const { return filters.empty() || (filters.size() == 1 && filters[0] == "PASS"); }

bool VcfRecord::is_indel() const
{
    int gt = called_allele();
    if (gt < 1 || (size_t)gt >= alleles.size()) return false;
    return alleles[0].size() != alleles[(size_t)gt].size();
}

std::string VcfRecord::slice(int64_t start, int64_t stop, int ix) const
{
    size_t gt;
    if (ix < 0) {
        int c = called_allele();
        gt = c < 0 ? 0 : (size_t)c;
    } else if ((size_t)ix < alleles.size()) gt = (size_t)ix;
    else return "";
    if (gt >= alleles.size()) return "";
    const std::string& al = alleles[gt];
    int64_t a0 = pos, a1 = pos + (int64_t)al.size();
    if (start >= a1 || a0 >= stop) return "";
    int64_t s = std::max(a0, start), e = std::min(a1, stop);
    size_t off = (size_t)(s - pos);
    size_t len = std::min((size_t)(e - s), al.size() - off);
    return al.substr(off, len);
}

int VcfRecord::argmatch(const VcfRecord& other) const
{
    int called = called_allele();
    int64_t called_len;
    if (called == 0) called_len = rlen();
    else if (called > 0) called_len = (int64_t)alleles[(size_t)called].size();
    else return -1;
    int64_t called_diff = std::llabs(called_len - rlen());
    int match_ix = -1;
    bool have_diff = false;
    int64_t match_diff = 0;
    const int64_t oiv0 = pos, oiv1 = pos + called_len;
    std::string other_ref = other.slice(pos, INT64_MAX, 0);
    for (size_t i = 0; i < other.alleles.size(); ++i) {
        const std::string& al = other.alleles[i];
        bool indel = al.size() != other.alleles[0].size();
        if (is_indel() != indel) continue;
        std::string seq = slice(other.pos, other.pos + (int64_t)al.size(), -1);
        if (seq.empty()) continue;
        std::string other_seq = other.slice(oiv0, oiv1, (int)i);
        int64_t diff = std::llabs((int64_t)other_ref.size() - (int64_t)al.size());
        if (seq != other_seq) continue;
        if (called == 0 && i == 0) return 0;
        if (!is_indel() && !indel) {
            int64_t ov0 = std::max(pos, other.pos), ov1 = std::min(end(), other.end());
            int64_t ro0 = ov1, ro1 = std::max(end(), other.end());
            int64_t lo0 = std::min(pos, other.pos), lo1 = ov0;
            std::string self_overlap = slice(ov0, ov1, -1);
            std::string self_left = pos == lo0 ? slice(lo0, lo1, -1) : other.slice(lo0, lo1, 0);
            std::string self_right = end() == ro1 ? slice(ro0, ro1, -1) : other.slice(ro0, ro1, 0);
            std::string other_overlap = other.slice(ov0, ov1, (int)i);
            std::string other_left = other.pos == lo0 ? other.slice(lo0, lo1, (int)i) : slice(lo0, lo1, 0);
            std::string other_right = other.end() == ro1 ? other.slice(ro0, ro1, (int)i) : slice(ro0, ro1, 0);
            if (other_left + other_overlap + other_right != self_left + self_overlap + self_right) continue;
        }
        int64_t diff_diff = std::llabs(called_diff - diff);
        if (!(have_diff && match_diff <= diff_diff)) {
            match_diff = diff_diff;
            have_diff = true;
            match_ix = (int)i;
        }
    }
    return match_ix;
}

std::string VcfRecord::to_line() const
{
    std::ostringstream o;
    o << chrom << "\t" << pos + 1 << "\t" << id << "\t" << alleles[0] << "\t";
    if (alleles.size() == 1) o << ".";
    for (size_t i = 1; i < alleles.size(); ++i) o << (i > 1 ? "," : "") << alleles[i];
    o << "\t" << qual << "\t";
    if (filters.empty()) o << ".";
    for (size_t i = 0; i < filters.size(); ++i) o << (i ? ";" : "") << filters[i];
    o << "\t";
    if (info.empty()) o << ".";
    for (size_t i = 0; i < info.size(); ++i) {
        o << (i ? ";" : "") << info[i].first;
        if (!info[i].second.empty()) o << "=" << info[i].second;
    }
    if (!format.empty()) {
        o << "\t";
        for (size_t i = 0; i < format.size(); ++i) o << (i ? ":" : "") << format[i];
        o << "\t";
        for (size_t i = 0; i < sample.size(); ++i) o << (i ? ":" : "") << sample[i];
    }
    return o.str();
}

VcfRecord parse_vcf_line(const std::string& line)
{
    std::vector<std::string> t = split(line, '\t');
    if (t.size() < 8) throw Error(DRPRG_EFORMAT, "VCF record with fewer than 8 columns: " + line.substr(0, 60));
    VcfRecord r;
    r.chrom = t[0];
    r.pos = std::atoll(t[1].c_str()) - 1;
    r.id = t[2];
    r.alleles.push_back(t[3]);
    if (t[4] != ".")
        for (auto& a : split(t[4], ',')) r.alleles.push_back(a);
    r.qual = t[5];
    if (t[6] != ".") r.filters = split(t[6], ';');
    if (t[7] != ".")
        for (auto& kv : split(t[7], ';')) {
            size_t e = kv.find('=');
            if (e == std::string::npos) r.info.emplace_back(kv, "");
            else r.info.emplace_back(kv.substr(0, e), kv.substr(e + 1));
        }
    if (t.size() > 9) {
        r.format = split(t[8], ':');
        r.sample = split(t[9], ':');
    }
    return r;
}

VcfFile read_vcf(const std::string& path)
{
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) throw Error(DRPRG_ENOENT, "cannot open VCF " + path);
    std::string text;
    char buf[1 << 16];
    int n;
    while ((n = gzread(fp, buf, sizeof buf)) > 0) text.append(buf, (size_t)n);
    gzclose(fp);
    VcfFile f;
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line.compare(0, 2, "##") == 0) f.header.push_back(line);
        else if (line[0] == '#') f.column_line = line;
        else f.records.push_back(parse_vcf_line(line));
    }
    if (f.column_line.empty()) throw Error(DRPRG_EFORMAT, path + " has no #CHROM line");
    return f;
}

std::vector<std::string> VcfFile::contigs() const
{
    std::vector<std::string> out;
    for (const std::string& h : header)
        if (h.compare(0, 13, "##contig=<ID=") == 0) {
            size_t e = h.find_first_of(",>", 13);
            out.push_back(h.substr(13, e - 13));
        }
    return out;
}

// ---- BGZF + BCF2 --------------------------------------------------------------------------------
static std::string bgzf_inflate_all(const std::string& path)
{
    // BGZF is a series of gzip members; zlib's gz* API reads concatenated members transparently
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) throw Error(DRPRG_ENOENT, "cannot open " + path);
    std::string out;
    char buf[1 << 16];
    int n;
    while ((n = gzread(fp, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
    gzclose(fp);
    return out;
}

namespace {
struct Cursor {
    const unsigned char* p;
    const unsigned char* end;
    template <typename T> T get()
    {
        if (p + sizeof(T) > end) throw Error(DRPRG_EFORMAT, "truncated BCF");
        T v;
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    // typed descriptor -> (type, count)
    void typed(int& type, int& count)
    {
        uint8_t b = get<uint8_t>();
        type = b & 0xF;
        count = b >> 4;
        if (count == 15) {
            int t2, c2;
            typed(t2, c2);
            count = (int)read_int(t2);
        }
    }
    int64_t read_int(int type)
    {
        switch (type) {
        case 1: return get<int8_t>();
        case 2: return get<int16_t>();
        case 3: return get<int32_t>();
        default: throw Error(DRPRG_EFORMAT, "BCF: expected an integer type");
        }
    }
    std::string read_string()
    {
        int type, count;
        typed(type, count);
        if (type != 7 && count != 0) throw Error(DRPRG_EFORMAT, "BCF: expected a string");
        if (p + count > end) throw Error(DRPRG_EFORMAT, "truncated BCF");
        std::string s((const char*)p, (size_t)count);
        p += count;
        while (!s.empty() && s.back() == '\0') s.pop_back();
        return s;
    }
    void skip_value(int type, int count)
    {
        static const int size[16] = { 0, 1, 2, 4, 0, 4, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0 };
        p += (size_t)size[type] * (size_t)count;
        if (p > end) throw Error(DRPRG_EFORMAT, "truncated BCF");
    }
};
} // namespace

std::vector<PanelRecordBcf> read_panel_bcf(const std::string& path)
{
    const std::string raw = bgzf_inflate_all(path);
    Cursor c { (const unsigned char*)raw.data(), (const unsigned char*)raw.data() + raw.size() };
    if (raw.size() < 9 || std::memcmp(raw.data(), "BCF\2", 4) != 0) throw Error(DRPRG_EFORMAT, path + " is not a BCF2 file");
    c.p += 5;
    uint32_t l_text = c.get<uint32_t>();
    std::string text((const char*)c.p, l_text);
    c.p += l_text;
    // dictionaries: contigs in order of appearance; strings (FILTER/INFO/FORMAT ids) with PASS = 0 or explicit IDX
    std::vector<std::string> contigs;
    std::map<int, std::string> strings;
    strings[0] = "PASS";
    int next_str = 1;
    std::set<std::string> seen { "PASS" };
    std::istringstream hs(text);
    std::string line;
    auto attr = [](const std::string& l, const std::string& key) -> std::string {
        size_t p = l.find(key + "=");
        if (p == std::string::npos) return "";
        p += key.size() + 1;
        size_t e = l.find_first_of(",>", p);
        return l.substr(p, e - p);
    };
    while (std::getline(hs, line)) {
        if (line.compare(0, 9, "##contig=") == 0) contigs.push_back(attr(line, "ID"));
        else if (line.compare(0, 9, "##FILTER=") == 0 || line.compare(0, 7, "##INFO=") == 0 || line.compare(0, 9, "##FORMAT=") == 0) {
            std::string id = attr(line, "ID"), idx = attr(line, "IDX");
            if (!idx.empty()) {
                strings[std::atoi(idx.c_str())] = id;
                seen.insert(id);
            } else if (!seen.count(id)) {
                seen.insert(id);
                strings[next_str++] = id;
            }
        }
    }
    std::vector<PanelRecordBcf> out;
    while (c.p + 8 <= c.end) {
        uint32_t l_shared = c.get<uint32_t>(), l_indiv = c.get<uint32_t>();
        Cursor s { c.p, c.p + l_shared };
        c.p += (size_t)l_shared + l_indiv;
        if (c.p > c.end) throw Error(DRPRG_EFORMAT, "truncated BCF record");
        int32_t chrom = s.get<int32_t>(), pos = s.get<int32_t>();
        (void)s.get<int32_t>(); // rlen
        (void)s.get<float>();   // qual
        uint32_t n_allele_info = s.get<uint32_t>();
        (void)s.get<uint32_t>(); // n_fmt_sample
        uint32_t n_info = n_allele_info & 0xFFFF, n_allele = n_allele_info >> 16;
        PanelRecordBcf r;
        if (chrom < 0 || (size_t)chrom >= contigs.size()) throw Error(DRPRG_EFORMAT, "BCF: contig id out of range");
        r.rec.chrom = contigs[(size_t)chrom];
        r.rec.pos = pos;
        r.rec.id = s.read_string();
        for (uint32_t i = 0; i < n_allele; ++i) r.rec.alleles.push_back(s.read_string());
        {
            int type, count;
            s.typed(type, count);
            s.skip_value(type, count); // FILTER
        }
        for (uint32_t i = 0; i < n_info; ++i) {
            int kt, kc;
            s.typed(kt, kc);
            int key = (int)s.read_int(kt);
            int type, count;
            s.typed(type, count);
            const std::string& name = strings[key];
            if (type == 7) {
                std::string v((const char*)s.p, (size_t)count);
                s.p += count;
                while (!v.empty() && v.back() == '\0') v.pop_back();
                if (name == "DRUGS") r.drugs = split(v, ',');
                else if (name == "RES") r.residue = v;
            } else {
                s.skip_value(type, count);
            }
        }
        out.push_back(std::move(r));
    }
    return out;
}

} // namespace report
} // namespace drprg
