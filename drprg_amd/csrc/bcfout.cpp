// bcfout.cpp -- BCF2.2 writer (BGZF container) for the annotated VCF of `drprg predict`.
//
// The reference writes <sample>.drprg.bcf through rust-htslib (/root/reference/src/predict.rs:429-431, Format::Bcf); this
// build writes the same records as VCF text first (report::annotate_vcf) and converts here, so that the file name and format
// of the report surface are the reference's.  Layout per the VCF/BCF specification (hts-specs VCFv4.3 section 6): magic
// "BCF\2\2", l_text, the NUL-terminated header text, then per record l_shared / l_indiv and the typed values; FILTER / INFO /
// FORMAT keys are indexes into the dictionary of header IDs in order of first appearance with PASS = 0, CHROM into the
// ##contig lines.  Checked by an independent Python decoder (tests/bcf_decode.py, itself pinned on the reference's own
// htslib-written panel.bcf).
#include "report.h"
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <zlib.h>

namespace drprg {
namespace report {

namespace {

struct FieldDef {
    int idx = -1;
    std::string number, type; // INFO / FORMAT only
};

struct Dict {
    std::map<std::string, int> contig;
    std::map<std::string, FieldDef> filter, info, format;
};

std::string attr(const std::string& l, const std::string& key)
{
    size_t p = l.find(key + "=");
    if (p == std::string::npos) return "";
    p += key.size() + 1;
    size_t e = l.find_first_of(",>", p);
    return l.substr(p, e - p);
}

// dictionary of strings: IDs of FILTER / INFO / FORMAT lines in order of first appearance, PASS first
Dict build_dict(std::vector<std::string>& header)
{
    bool has_pass = false;
    for (const std::string& l : header)
        if (l.compare(0, 9, "##FILTER=") == 0 && attr(l, "ID") == "PASS") has_pass = true;
    if (!has_pass) { // htslib adds it; it must be entry 0 of the dictionary
        size_t at = 0;
        while (at < header.size() && header[at].compare(0, 13, "##fileformat=") == 0) ++at;
        header.insert(header.begin() + (long)at, "##FILTER=<ID=PASS,Description=\"All filters passed\">");
    }
    Dict d;
    std::map<std::string, int> strings;
    strings["PASS"] = 0;
    int next = 1;
    for (const std::string& l : header) {
        if (l.compare(0, 9, "##contig=") == 0) {
            const std::string id = attr(l, "ID");
            if (!d.contig.count(id)) {
                const int n = (int)d.contig.size();
                d.contig[id] = n;
            }
            continue;
        }
        std::map<std::string, FieldDef>* which = nullptr;
        if (l.compare(0, 9, "##FILTER=") == 0) which = &d.filter;
        else if (l.compare(0, 7, "##INFO=") == 0) which = &d.info;
        else if (l.compare(0, 9, "##FORMAT=") == 0) which = &d.format;
        if (!which) continue;
        const std::string id = attr(l, "ID");
        if (!strings.count(id)) strings[id] = next++;
        FieldDef f;
        f.idx = strings[id];
        f.number = attr(l, "Number");
        f.type = attr(l, "Type");
        (*which)[id] = f;
    }
    return d;
}

struct Buf {
    std::string b;
    template <typename T> void put(T v) { b.append(reinterpret_cast<const char*>(&v), sizeof(T)); }
    void descriptor(size_t count, int type)
    {
        if (count < 15) put<uint8_t>((uint8_t)((count << 4) | (unsigned)type));
        else {
            put<uint8_t>((uint8_t)(0xF0 | (unsigned)type));
            typed_int((int64_t)count);
        }
    }
    static int int_type(int64_t lo, int64_t hi)
    {
        if (lo >= -120 && hi <= 127) return 1;
        if (lo >= -32760 && hi <= 32767) return 2;
        return 3;
    }
    void raw_int(int64_t v, int type)
    {
        if (type == 1) put<int8_t>((int8_t)v);
        else if (type == 2) put<int16_t>((int16_t)v);
        else put<int32_t>((int32_t)v);
    }
    void typed_int(int64_t v)
    {
        const int t = int_type(v, v);
        put<uint8_t>((uint8_t)((1u << 4) | (unsigned)t));
        raw_int(v, t);
    }
    void typed_string(const std::string& s)
    {
        descriptor(s.size(), 7);
        b += s;
    }
    // vector of integers, "." = missing
    void typed_ints(const std::vector<std::string>& vals)
    {
        std::vector<int64_t> v;
        std::vector<bool> missing;
        int64_t lo = 0, hi = 0;
        for (const std::string& s : vals) {
            const bool m = s == "." || s.empty();
            missing.push_back(m);
            const int64_t x = m ? 0 : std::strtoll(s.c_str(), nullptr, 10);
            if (x < INT32_MIN + 8 || x > INT32_MAX) throw Error(DRPRG_EFORMAT, "BCF writer: integer " + s + " does not fit 32 bits"); // (the 8 lowest are reserved)
            v.push_back(x);
            lo = std::min(lo, x);
            hi = std::max(hi, x);
        }
        const int t = int_type(lo, hi);
        descriptor(v.size(), t);
        for (size_t i = 0; i < v.size(); ++i) {
            if (missing[i]) raw_int(t == 1 ? INT8_MIN : t == 2 ? INT16_MIN : INT32_MIN, t);
            else raw_int(v[i], t);
        }
    }
    void typed_floats(const std::vector<std::string>& vals)
    {
        descriptor(vals.size(), 5);
        for (const std::string& s : vals) {
            if (s == "." || s.empty()) put<uint32_t>(0x7F800001u); // missing
            else put<float>(std::strtof(s.c_str(), nullptr));
        }
    }
};

std::vector<std::string> split_commas(const std::string& s)
{
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        size_t e = s.find(',', a);
        out.push_back(s.substr(a, e == std::string::npos ? std::string::npos : e - a));
        if (e == std::string::npos) break;
        a = e + 1;
    }
    return out;
}

void typed_value(Buf& o, const FieldDef& f, const std::string& value, const std::string& what)
{
    if (f.type == "Flag") {
        o.put<uint8_t>(0x00); // no value
    } else if (f.type == "Integer") {
        o.typed_ints(split_commas(value));
    } else if (f.type == "Float") {
        o.typed_floats(split_commas(value));
    } else if (f.type == "String" || f.type == "Character") {
        o.typed_string(value);
    } else {
        throw Error(DRPRG_EFORMAT, "BCF writer: header line of " + what + " has no usable Type");
    }
}

void encode_record(const VcfRecord& r, const Dict& d, Buf& out)
{
    Buf sh, ind;
    auto ci = d.contig.find(r.chrom);
    if (ci == d.contig.end()) throw Error(DRPRG_EFORMAT, "BCF writer: contig " + r.chrom + " is not in the header");
    sh.put<int32_t>(ci->second);
    sh.put<int32_t>((int32_t)r.pos);
    sh.put<int32_t>((int32_t)r.rlen());
    if (r.qual == "." || r.qual.empty()) sh.put<uint32_t>(0x7F800001u);
    else sh.put<float>(std::strtof(r.qual.c_str(), nullptr));
    uint32_t n_info = 0;
    for (const auto& kv : r.info) n_info += !kv.first.empty(); // (a stray ';' in the text leaves an empty entry)
    sh.put<uint32_t>(((uint32_t)r.alleles.size() << 16) | n_info);
    sh.put<uint32_t>(((uint32_t)r.format.size() << 24) | 1u); // one sample
    sh.typed_string(r.id == "." ? std::string() : r.id);
    for (const std::string& a : r.alleles) sh.typed_string(a);
    { // FILTER: vector of dictionary indexes ("." = no value)
        std::vector<std::string> ids;
        for (const std::string& f : r.filters) {
            auto it = f == "PASS" ? d.filter.end() : d.filter.find(f);
            if (f == "PASS") ids.push_back("0");
            else if (it == d.filter.end()) throw Error(DRPRG_EFORMAT, "BCF writer: FILTER " + f + " is not in the header");
            else ids.push_back(std::to_string(it->second.idx));
        }
        if (ids.empty()) sh.put<uint8_t>(0x00);
        else sh.typed_ints(ids);
    }
    for (const auto& kv : r.info) {
        if (kv.first.empty()) continue;
        auto it = d.info.find(kv.first);
        if (it == d.info.end()) throw Error(DRPRG_EFORMAT, "BCF writer: INFO " + kv.first + " is not in the header");
        sh.typed_int(it->second.idx);
        typed_value(sh, it->second, kv.second, "INFO " + kv.first);
    }
    for (size_t i = 0; i < r.format.size(); ++i) {
        auto it = d.format.find(r.format[i]);
        if (it == d.format.end()) throw Error(DRPRG_EFORMAT, "BCF writer: FORMAT " + r.format[i] + " is not in the header");
        ind.typed_int(it->second.idx);
        const std::string& v = i < r.sample.size() ? r.sample[i] : std::string(".");
        if (r.format[i] == "GT") { // (allele + 1) << 1 | phased per allele; "." = 0
            // (the width follows the largest value, as htslib's does: allele 63 of a very multi-allelic site no longer fits int8)
            std::vector<int64_t> g;
            size_t a = 0;
            bool phased = false;
            int64_t hi = 0;
            while (a <= v.size()) {
                size_t e = v.find_first_of("/|", a);
                const std::string tok = v.substr(a, e == std::string::npos ? std::string::npos : e - a);
                const int64_t allele = (tok == "." || tok.empty()) ? -1 : std::strtoll(tok.c_str(), nullptr, 10);
                if (allele < -1 || allele > (INT32_MAX >> 1) - 1) throw Error(DRPRG_EFORMAT, "BCF writer: GT " + v + " is out of range");
                g.push_back(((allele + 1) << 1) | (phased ? 1 : 0));
                hi = std::max(hi, g.back());
                if (e == std::string::npos) break;
                phased = v[e] == '|';
                a = e + 1;
            }
            const int t = Buf::int_type(0, hi);
            ind.descriptor(g.size(), t);
            for (int64_t x : g) ind.raw_int(x, t);
        } else {
            typed_value(ind, it->second, v, "FORMAT " + r.format[i]);
        }
    }
    out.put<uint32_t>((uint32_t)sh.b.size());
    out.put<uint32_t>((uint32_t)ind.b.size());
    out.b += sh.b;
    out.b += ind.b;
}

// BGZF: gzip members of <= 64 KB of input each, 'BC' extra field = member size - 1, closed by the empty EOF member
void write_bgzf(const std::string& path, const std::string& data)
{
    std::ofstream o(path, std::ios::binary);
    if (!o) throw Error(DRPRG_EIO, "cannot write " + path);
    auto member = [&](const char* p, size_t n) {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw Error(DRPRG_EIO, "deflateInit2 failed");
        std::string body(deflateBound(&zs, (uLong)n) + 16, '\0');
        zs.next_in = (Bytef*)p;
        zs.avail_in = (uInt)n;
        zs.next_out = (Bytef*)&body[0];
        zs.avail_out = (uInt)body.size();
        const int rc = deflate(&zs, Z_FINISH);
        const size_t blen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) throw Error(DRPRG_EIO, "deflate failed");
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)p, (uInt)n);
        const uint16_t bsize = (uint16_t)(12 + 6 + blen + 8 - 1);
        const unsigned char head[18] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 0xff), (unsigned char)(bsize >> 8) };
        o.write((const char*)head, 18);
        o.write(body.data(), (std::streamsize)blen);
        const uint32_t tail[2] = { crc, (uint32_t)n };
        o.write((const char*)tail, 8);
    };
    constexpr size_t BLOCK = 0xff00; // htslib's block size
    for (size_t off = 0; off < data.size(); off += BLOCK) member(data.data() + off, std::min(BLOCK, data.size() - off));
    member(nullptr, 0);
    if (!o) throw Error(DRPRG_EIO, "short write to " + path);
}

} // namespace

void write_bcf(const std::string& path, const VcfFile& vcf_in)
{
    VcfFile vcf = vcf_in;
    const Dict d = build_dict(vcf.header);
    std::string text;
    for (const std::string& l : vcf.header) text += l + "\n";
    text += vcf.column_line + "\n";
    Buf out;
    out.b = std::string("BCF\2\2", 5);
    out.put<uint32_t>((uint32_t)text.size() + 1);
    out.b += text;
    out.b.push_back('\0');
    for (const VcfRecord& r : vcf.records) encode_record(r, d, out);
    write_bgzf(path, out.b);
}

void vcf_to_bcf(const std::string& vcf_path, const std::string& bcf_path) { write_bcf(bcf_path, read_vcf(vcf_path)); }

} // namespace report
} // namespace drprg
