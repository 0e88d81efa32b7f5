// ingest.h -- multi-threaded FASTA/FASTQ ingest for the CLI path (NEXT-4).
//
// The kernels map 10 M reads in under 2 ms; end to end the path is bound by text parsing and the host->device copy.
// A plain file is memory-mapped and cut at record boundaries into slices that worker threads parse straight into
// pinned buffers (so the H2D copy runs at DMA speed); gzip input is inflated window by window -- BGZF members in
// parallel with libdeflate, a plain gzip stream by all threads at once (pgunzip.h), a small one in one libdeflate call, anything else by zlib -- and cut the same way.  Read order is irrelevant to the result (coverage is a sum), so workers submit batches independently.
#pragma once
#include "common.h"
#include <functional>

namespace drprg {

struct PinnedBatch {
    uint8_t* bases = nullptr;    // ASCII; packed: the 2-bit words (pack.h), 16 bases per u32
    uint64_t* offsets = nullptr; // offsets[0] == 0 (in bases, whatever the format)
    uint64_t n_reads = 0, n_bases = 0;
    bool packed = false;
    const uint64_t* npos = nullptr; // packed: ascending positions of the bases that are not ACGTacgt (pageable memory)
    uint64_t n_npos = 0;
};

struct IngestHooks {
    std::function<void*(size_t)> alloc;            // pinned allocation (falls back to malloc when null)
    std::function<void(void*)> release;
    std::function<void(const PinnedBatch&)> submit; // called by one thread at a time ...
    bool concurrent_submit = false;                 // ... unless set: then by any parser thread, and submit does its own locking
    bool packed = false;                            // the parser threads pack the bases to 2 bits as they copy them (pack.h): a block is a
                                                    // quarter of the bytes to page-lock and to move over PCIe
};

struct IngestStats {
    uint64_t reads = 0, bases = 0, batches = 0;
    bool parallel = false;
    int gz_mode = 0; // 0 plain text, 1 BGZF (members inflated in parallel), 2 one gzip member in one libdeflate call, 3 zlib streaming,
                     // 4 one plain gzip stream inflated by all threads (pgunzip.h)
};

// Parses `path` (fasta/fastq, plain or .gz) with `threads` parser threads and feeds every batch to hooks.submit.
// Throws Error on malformed input.
IngestStats ingest_fastx(const std::string& path, int threads, const IngestHooks& hooks);

} // namespace drprg
