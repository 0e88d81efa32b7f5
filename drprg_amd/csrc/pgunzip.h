// pgunzip.h -- one plain gzip stream inflated by many threads (NEXT-4: `.fq.gz` is the reference's
// normal input, /root/reference/docs/src/guide/predict.md:25-36, and one zlib / libdeflate thread is three orders of
// magnitude under the kernels).
//
// A deflate stream has no index, but it can be entered at any block boundary if the 32 KB of text before it are treated as
// unknown: the compressed file is cut into chunks, every chunk's thread looks for the first dynamic-Huffman block header
// at or after its offset (bit by bit: header fields in range, complete code-length / literal / distance codes, and the
// block decodes) and inflates from there into 16-bit symbols -- a byte, or 0x8000 + i for "byte i of the 32 KB before my
// start" -- until it reaches the first dynamic block at or after the next chunk's offset.  The chunks are then stitched in
// file order: a chunk is accepted iff it starts exactly where its predecessor ended (anything else -- a header look-alike,
// no dynamic block in range, an over-long chunk -- is inflated again from the known position); the last 32 KB of every
// accepted chunk, resolved, are the next one's window, and with the windows known all symbols are resolved to bytes by all
// threads.  CRC-32 and length of every member are checked like gzip does.  (The approach of pugz / rapidgzip; own code.)
#pragma once
#include "common.h"
#include <memory>

namespace drprg {

class ParallelGunzip {
public:
    // data/len: the whole compressed file (stays mapped for the lifetime of the object); chunk_bytes 0 = automatic
    ParallelGunzip(const unsigned char* data, size_t len, int threads, size_t chunk_bytes = 0);
    ~ParallelGunzip();
    // up to cap bytes of text into dst; 0 = end of the stream.  Throws Error(DRPRG_EIO) on a corrupt stream.
    size_t read(char* dst, size_t cap);
    // chunks accepted as their threads inflated them / inflated again from the known position (diagnostics, tests)
    uint64_t chunks_accepted() const;
    uint64_t chunks_redone() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

} // namespace drprg
