// fastx.cpp -- see fastx.h
#include "fastx.h"
#include <cctype>
#include <cstring>

namespace drprg {

FastxReader::FastxReader(const std::string& path) : path_(path)
{
    fp_ = gzopen(path.c_str(), "rb"); // transparently reads plain files too
    if (!fp_) throw Error(DRPRG_ENOENT, "cannot open reads file " + path);
    gzbuffer(fp_, 1 << 20);
    buf_.resize(1 << 22);
}

FastxReader::~FastxReader()
{
    if (fp_) gzclose(fp_);
}

bool FastxReader::fill()
{
    if (eof_) return false;
    int n = gzread(fp_, buf_.data(), (unsigned)buf_.size());
    if (n < 0) throw Error(DRPRG_EIO, "read error in " + path_);
    pos_ = 0;
    len_ = (size_t)n;
    if (n == 0) eof_ = true;
    return n > 0;
}

bool FastxReader::getline_(std::string& s)
{
    s.clear();
    bool got = false;
    while (true) {
        if (pos_ == len_ && !fill()) break;
        got = true;
        const unsigned char* b = buf_.data() + pos_;
        const unsigned char* e = (const unsigned char*)memchr(b, '\n', len_ - pos_);
        if (e) {
            s.append((const char*)b, (size_t)(e - b));
            pos_ += (size_t)(e - b) + 1;
            break;
        }
        s.append((const char*)b, len_ - pos_);
        pos_ = len_;
    }
    if (!s.empty() && s.back() == '\r') s.pop_back();
    return got;
}

bool FastxReader::next_batch(ReadBatch& out, uint64_t max_reads, uint64_t max_bases, bool keep_names)
{
    out.clear();
    std::string header, seq, plus, qual;
    while (out.n_reads() < max_reads && out.bases.size() < max_bases) {
        if (have_pending_) {
            header = pending_header_;
            have_pending_ = false;
        } else {
            bool any = false;
            while (getline_(header)) {
                if (!header.empty()) { any = true; break; }
            }
            if (!any) break;
        }
        if (header[0] == '@') { // FASTQ: 4-line records (multi-line sequence tolerated)
            seq.clear();
            while (getline_(line_)) {
                if (!line_.empty() && line_[0] == '+') break;
                seq += line_;
            }
            size_t need = seq.size(), gotq = 0;
            while (gotq < need && getline_(line_)) gotq += line_.size();
            if (gotq < need) throw Error(DRPRG_EFORMAT, "truncated FASTQ record in " + path_);
        } else if (header[0] == '>') {
            seq.clear();
            while (getline_(line_)) {
                if (!line_.empty() && line_[0] == '>') {
                    pending_header_ = line_;
                    have_pending_ = true;
                    break;
                }
                seq += line_;
            }
        } else {
            throw Error(DRPRG_EFORMAT, "not a FASTA/FASTQ record in " + path_ + ": " + header.substr(0, 40));
        }
        out.bases.insert(out.bases.end(), seq.begin(), seq.end());
        out.offsets.push_back(out.bases.size());
        if (keep_names) {
            size_t e = header.find_first_of(" \t");
            out.names.push_back(header.substr(1, e == std::string::npos ? std::string::npos : e - 1));
        }
    }
    return out.n_reads() > 0;
}

std::vector<std::pair<std::string, std::string>> read_fasta(const std::string& path)
{
    FastxReader r(path);
    ReadBatch b;
    std::vector<std::pair<std::string, std::string>> out;
    while (r.next_batch(b, 1 << 20, 1ull << 32, true)) {
        for (uint64_t i = 0; i < b.n_reads(); ++i) {
            std::string s(b.bases.begin() + (long)b.offsets[i], b.bases.begin() + (long)b.offsets[i + 1]);
            for (char& c : s) c = (char)std::toupper((unsigned char)c);
            out.emplace_back(b.names[i], s);
        }
    }
    return out;
}

} // namespace drprg
