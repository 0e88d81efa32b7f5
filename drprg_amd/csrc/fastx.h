// fastx.h -- FASTA/FASTQ reader (plain or gzip) producing batches in the layout the kernels consume:
// concatenated bases + u64 offsets.  drprg accepts fasta/fastq, gz or plain
// (/root/reference/src/predict.rs:166-170).
#pragma once
#include "common.h"
#include <zlib.h>

namespace drprg {

struct ReadBatch {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets { 0 };
    std::vector<std::string> names; // filled only when keep_names
    uint64_t n_reads() const { return offsets.size() - 1; }
    void clear()
    {
        bases.clear();
        offsets.assign(1, 0);
        names.clear();
    }
};

class FastxReader {
public:
    explicit FastxReader(const std::string& path);
    ~FastxReader();
    // append up to max_reads / max_bases to `out` (cleared first); false when the file is exhausted
    bool next_batch(ReadBatch& out, uint64_t max_reads, uint64_t max_bases, bool keep_names = false);

private:
    bool fill();
    int getc_();
    bool getline_(std::string& s);
    gzFile fp_ = nullptr;
    std::vector<unsigned char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    std::string path_, line_, pending_header_;
    bool have_pending_ = false;
};

// whole-file convenience (genes.fa): name -> sequence (upper-cased), in file order
std::vector<std::pair<std::string, std::string>> read_fasta(const std::string& path);

} // namespace drprg
