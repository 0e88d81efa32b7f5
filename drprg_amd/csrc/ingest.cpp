// ingest.cpp -- see ingest.h
#include "ingest.h"
#include "pack.h"
#include "pgunzip.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <dlfcn.h>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace drprg {

namespace {

// The text is cut into slices, one parser task each; a parser thread copies the bases of its slices into its page-locked block
// and hands the block over when it is full (one H2D copy + one launch sequence per block).  Page-locking costs time per byte
// (Mapper::pinned_alloc), hence small blocks, allocated only by workers that get work.
constexpr size_t SLICE_BYTES = 32u << 20; // text handed to one parser task (compressed input: a window of inflated text)
// ... of a plain-text file read with pread: smaller (8 MB), because the parser thread keeps a buffer of 1.5 slices, every page of
// which is faulted in once and torn down at exit (DRPRG_INGEST_SLICE_MB; `drprg predict` on 10 M reads, medians of 7 runs on one
// box: 0.41 s with 32 MB slices and 24 MB blocks, 0.25 s with 8 and 12 -- profiles/r03/e2e_cli.txt)
inline size_t plain_slice_bytes()
{
    static const size_t n = [] {
        const char* e = std::getenv("DRPRG_INGEST_SLICE_MB");
        const long mb = e ? std::atol(e) : 0;
        return (size_t)(mb >= 1 && mb <= 1024 ? mb : 8) << 20;
    }();
    return n;
}
// pinned block: bases capacity (12 MB; DRPRG_INGEST_BLOCK_MB; also the longest read the parallel ingest takes) ...
inline size_t block_bases_now()
{
    static const size_t n = [] {
        const char* e = std::getenv("DRPRG_INGEST_BLOCK_MB");
        const long mb = e ? std::atol(e) : 0;
        return (size_t)(mb >= 1 && mb <= 1024 ? mb : 12) << 20;
    }();
    return n;
}
#define BLOCK_BASES (block_bases_now())
#define BLOCK_READS (BLOCK_BASES / 96) // ... and read capacity (flushed early when either fills up: reads under 96 bases)

struct Block {
    uint8_t* bases = nullptr; // ASCII, or (packed) the 2-bit words, 8-byte aligned
    uint64_t* offsets = nullptr;
    uint64_t n_reads = 0, n_bases = 0;
    bool packed = false;
    std::vector<uint64_t> npos; // packed: positions of the bases that are not ACGTacgt
    // the bases of one read (or one line of a multi-line record) behind what the block holds
    inline void append(const char* seq, size_t len)
    {
        if (packed) {
            pack_append(reinterpret_cast<uint64_t*>(bases), n_bases, seq, len, npos);
        } else {
            std::memcpy(bases + n_bases, seq, len);
            n_bases += len;
        }
    }
    // the block is handed over once it holds this many bases: BLOCK_BASES, except for a worker's first hand-over, which comes
    // early and at a different fill for every worker -- otherwise all workers fill their first block at the same moment, the
    // copy engine idles until then and works through a burst afterwards (measured: 10 M x 150 bp, first copies at 67 of 95 ms)
    size_t flush_at = 0;
};

struct Shared {
    const IngestHooks& hooks;
    std::mutex submit_mu, err_mu;
    std::atomic<uint64_t> reads { 0 }, bases { 0 }, batches { 0 };
    std::atomic<uint32_t> first_flush { 0 };
    // DRPRG_INGEST_DEBUG=1: where the wall time of a call goes (nanoseconds)
    const bool debug = std::getenv("DRPRG_INGEST_DEBUG") != nullptr;
    std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();
    std::atomic<int64_t> ns_first_submit { -1 }, ns_submit_wait { 0 }, ns_submit_run { 0 }, ns_parse { 0 }, ns_alloc { 0 }, ns_read { 0 };
    int64_t now_ns() const { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_start).count(); }
    std::atomic<bool> failed { false };
    std::string error;
    int error_code = DRPRG_EFORMAT;

    explicit Shared(const IngestHooks& h) : hooks(h) {}
    void fail(int code, const std::string& m)
    {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.exchange(true)) {
            error = m;
            error_code = code;
        }
    }
    void submit(Block& b)
    {
        if (b.n_reads == 0) return;
        PinnedBatch pb { b.bases, b.offsets, b.n_reads, b.n_bases };
        pb.packed = b.packed;
        pb.npos = b.npos.data();
        pb.n_npos = b.npos.size();
        const int64_t t0 = debug ? now_ns() : 0;
        if (debug) {
            int64_t none = -1;
            ns_first_submit.compare_exchange_strong(none, t0);
        }
        if (hooks.concurrent_submit) hooks.submit(pb);
        else {
            std::lock_guard<std::mutex> g(submit_mu);
            const int64_t t1 = debug ? now_ns() : 0;
            hooks.submit(pb);
            if (debug) {
                ns_submit_wait += t1 - t0;
                ns_submit_run += now_ns() - t1;
            }
        }
        reads += b.n_reads;
        bases += b.n_bases;
        batches += 1;
        b.n_reads = 0;
        b.n_bases = 0;
        b.npos.clear();
        if (b.packed) reinterpret_cast<uint64_t*>(b.bases)[0] = 0; // (pack_append: a fresh stream)
        b.flush_at = BLOCK_BASES;
    }
    Block new_block()
    {
        Block b;
        b.packed = hooks.packed;
        const size_t bytes_b = (b.packed ? BLOCK_BASES / 4 + 16 : BLOCK_BASES) + 64, bytes_o = (BLOCK_READS + 2) * sizeof(uint64_t);
        b.bases = (uint8_t*)(hooks.alloc ? hooks.alloc(bytes_b) : std::malloc(bytes_b));
        b.offsets = (uint64_t*)(hooks.alloc ? hooks.alloc(bytes_o) : std::malloc(bytes_o));
        if (!b.bases || !b.offsets) throw Error(DRPRG_ENOMEM, "cannot allocate an ingest block");
        b.offsets[0] = 0;
        if (b.packed) reinterpret_cast<uint64_t*>(b.bases)[0] = 0;
        // first hand-over: between 1/16 and 16/16 of a block, a different sixteenth for consecutive workers
        b.flush_at = BLOCK_BASES / 16 * (1 + (size_t)(first_flush.fetch_add(1) % 16));
        return b;
    }
    void free_block(Block& b)
    {
        auto rel = [&](void* p) {
            if (!p) return;
            if (hooks.release) hooks.release(p);
            else std::free(p);
        };
        rel(b.bases);
        rel(b.offsets);
        b = Block();
    }
};

inline const char* find_nl(const char* p, const char* e) { return p < e ? (const char*)memchr(p, '\n', (size_t)(e - p)) : nullptr; }

// is the line starting at p the first line of a record?  (for FASTQ: an '@' line whose line after next starts with
// '+'; a quality line that starts with '@' is followed by a header and a sequence line, never by '+').
// Returns -1 when there is not enough text after p to decide.
int is_record_start(const char* p, const char* e, bool fastq)
{
    if (!fastq) return *p == '>' ? 1 : 0;
    if (*p != '@') return 0;
    const char* l1 = find_nl(p, e);
    const char* l2 = l1 ? find_nl(l1 + 1, e) : nullptr;
    if (!l2 || l2 + 1 >= e) return -1;
    return l2[1] == '+' ? 1 : 0;
}

// first record start at or after the first line start following p; e if none can be established
const char* next_record(const char* p, const char* e, bool fastq)
{
    const char* nl = find_nl(p, e);
    if (!nl) return e;
    p = nl + 1;
    while (p < e) {
        int r = is_record_start(p, e, fastq);
        if (r == 1) return p;
        if (r < 0) return e;
        nl = find_nl(p, e);
        if (!nl) return e;
        p = nl + 1;
    }
    return e;
}

inline void strip_cr(const char* s, const char*& e)
{
    if (e > s && e[-1] == '\r') --e;
}

// The records of a 4-line FASTQ slice, AVX2: the newline search is inline (32 bytes per compare; four glibc memchr calls per
// record cost more than the scanning they do: ~100 ns per 150-base record, i.e. 1 GB/s of text per thread, which made the parse
// -- not PCIe -- the limit of the end-to-end path).  Stops in front of the last 64 bytes of the slice (the vector loads must not
// run past a mapping's end) or at anything that is not the plain case; the caller's general loop continues from the returned
// position.
#if defined(__x86_64__)
// first newline at or after q, nullptr if none before `safe`
__attribute__((target("avx2"))) inline const char* find_nl_avx2(const char* q, const char* safe)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    while (q < safe) {
        const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)q), nl));
        if (m) {
            q += __builtin_ctz(m);
            return q < safe ? q : nullptr;
        }
        q += 32;
    }
    return nullptr;
}

__attribute__((target("avx2"))) const char* parse_fastq_avx2(const char* p, const char* e, Block& blk, Shared& sh)
{
    if (e - p < 128) return p;
    const char* const safe = e - 64;
    auto find = [&](const char* q) __attribute__((target("avx2"))) { return find_nl_avx2(q, safe); };
    while (p < safe) {
        if (*p != '@') return p; // blank line, CR, or malformed: the general loop decides
        const char* h_end = find(p);
        if (!h_end) return p;
        const char* seq = h_end + 1;
        const char* s_end = find(seq);
        if (!s_end || s_end + 1 >= safe || s_end[1] != '+') return p;
        const char* plus_end = find(s_end + 1);
        if (!plus_end) return p;
        const char* q_end = find(plus_end + 1);
        if (!q_end) return p;
        const char* s_stop = s_end;
        if (s_stop > seq && s_stop[-1] == '\r') --s_stop;
        const size_t len = (size_t)(s_stop - seq);
        if (len > BLOCK_BASES) return p;
        if (blk.n_bases + len > blk.flush_at || blk.n_reads + 1 > BLOCK_READS) sh.submit(blk);
        if (blk.packed) pack_append_overread_avx2(reinterpret_cast<uint64_t*>(blk.bases), blk.n_bases, seq, len, blk.npos); // (s_end + 64 < e: readable)
        else blk.append(seq, len);
        blk.offsets[++blk.n_reads] = blk.n_bases;
        p = q_end + 1;
    }
    return p;
}
#endif

// parse the records of [p, e) (whole records only) into the block, submitting whenever it fills up
void parse_slice(const char* p, const char* e, bool fastq, Block& blk, Shared& sh)
{
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && !std::getenv("DRPRG_PARSE_NO_SIMD");
    if (fastq && have_avx2) p = parse_fastq_avx2(p, e, blk, sh);
#endif
    while (p < e) {
        if (*p == '\n' || *p == '\r') {
            ++p;
            continue;
        }
        const char* h_end = find_nl(p, e);
        if (!h_end) throw Error(DRPRG_EFORMAT, "truncated record at the end of the reads file");
        const char* cursor = h_end + 1;
        if (fastq) {
            if (*p != '@') throw Error(DRPRG_EFORMAT, "not a 4-line FASTQ record");
            const char* s_end = find_nl(cursor, e);
            if (!s_end) throw Error(DRPRG_EFORMAT, "truncated FASTQ record");
            const char* plus = s_end + 1;
            if (plus >= e || *plus != '+') throw Error(DRPRG_EFORMAT, "not a 4-line FASTQ record");
            const char* plus_end = find_nl(plus, e);
            if (!plus_end) throw Error(DRPRG_EFORMAT, "truncated FASTQ record");
            const char* q_end = find_nl(plus_end + 1, e);
            const char* s_stop = s_end;
            strip_cr(cursor, s_stop);
            const size_t len = (size_t)(s_stop - cursor);
            if (len > BLOCK_BASES) throw Error(DRPRG_EOVERFLOW, "a read is longer than the ingest block");
            if (blk.n_bases + len > blk.flush_at || blk.n_reads + 1 > BLOCK_READS) sh.submit(blk);
            blk.append(cursor, len);
            blk.offsets[++blk.n_reads] = blk.n_bases;
            p = q_end ? q_end + 1 : e;
        } else {
            if (*p != '>') throw Error(DRPRG_EFORMAT, "not a FASTA record");
            // the record ends at the next '>' that starts a line
            const char* rec_end = cursor;
            while (rec_end < e && *rec_end != '>') {
                const char* nl = find_nl(rec_end, e);
                rec_end = nl ? nl + 1 : e;
            }
            const size_t bound = (size_t)(rec_end - cursor);
            if (bound > BLOCK_BASES) throw Error(DRPRG_EOVERFLOW, "a read is longer than the ingest block");
            if (blk.n_bases + bound > blk.flush_at || blk.n_reads + 1 > BLOCK_READS) sh.submit(blk);
            while (cursor < rec_end) {
                const char* nl = find_nl(cursor, rec_end);
                const char* stop = nl ? nl : rec_end;
                const char* next = nl ? nl + 1 : rec_end;
                strip_cr(cursor, stop);
                blk.append(cursor, (size_t)(stop - cursor));
                cursor = next;
            }
            blk.offsets[++blk.n_reads] = blk.n_bases;
            p = rec_end;
        }
    }
}

struct TextBuffer;
struct Slice {
    const char* begin = nullptr;
    const char* end = nullptr;
    std::shared_ptr<TextBuffer> owner; // inflated gzip text; null for a memory-mapped file
    std::shared_ptr<void> mapping;     // plain file: the slice's own mapping, unmapped by the parser thread when it is done with it
    uint64_t file_off = 0, file_len = 0; // plain file, begin == nullptr: the parser thread reads these bytes into a buffer of its own
};

class SliceQueue {
public:
    explicit SliceQueue(size_t cap) : cap_(cap) {}
    void push(Slice s)
    {
        std::unique_lock<std::mutex> l(mu_);
        not_full_.wait(l, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return;
        q_.push_back(std::move(s));
        not_empty_.notify_one();
    }
    bool pop(Slice& s)
    {
        std::unique_lock<std::mutex> l(mu_);
        not_empty_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        s = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> l(mu_);
        closed_ = true;
        not_empty_.notify_all();
        not_full_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable not_empty_, not_full_;
    std::deque<Slice> q_;
    size_t cap_;
    bool closed_ = false;
};

// Window buffers of the gzip path are recycled: a fresh 32 MB buffer per window means an mmap, 8 k page faults and an munmap
// (with its TLB shoot-downs on every parser thread) per window -- measured 4x slower with 8 threads than with 4.  They are raw
// (not value-initialised: a std::vector would write 32 MB of zeros first), 2 MB-aligned and advised as huge pages.
struct TextBuffer {
    char* p = nullptr;
    size_t n = 0;
    explicit TextBuffer(size_t size)
    {
        constexpr size_t HUGE = size_t(2) << 20;
        n = (size + HUGE - 1) / HUGE * HUGE;
        p = (char*)std::aligned_alloc(HUGE, n);
        if (!p) throw Error(DRPRG_EIO, "out of memory for a text window");
        madvise(p, n, MADV_HUGEPAGE);
    }
    ~TextBuffer() { std::free(p); }
    TextBuffer(const TextBuffer&) = delete;
    TextBuffer& operator=(const TextBuffer&) = delete;
    char* data() { return p; }
    size_t size() const { return n; }
};

class BufferPool {
public:
    std::shared_ptr<TextBuffer> acquire(size_t size)
    {
        std::unique_ptr<TextBuffer> v;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!free_.empty()) {
                v = std::move(free_.back());
                free_.pop_back();
            }
        }
        if (!v || v->size() < size) v.reset(new TextBuffer(size));
        TextBuffer* raw = v.release();
        return std::shared_ptr<TextBuffer>(raw, [this](TextBuffer* p) {
            std::unique_ptr<TextBuffer> own(p);
            std::lock_guard<std::mutex> g(mu_);
            if (free_.size() < 24) free_.push_back(std::move(own)); // (else freed here: at most ~1 GB of text buffers stays with the process)
        });
    }

private:
    std::mutex mu_;
    std::vector<std::unique_ptr<TextBuffer>> free_;
};

// multi-line FASTQ (sequence wrapped over several lines) is left to the serial reader
void require_four_line_fastq(const char* p, const char* e)
{
    while (p < e && (*p == '\n' || *p == '\r' || *p == ' ')) ++p;
    const char* l1 = find_nl(p, e);
    const char* l2 = l1 ? find_nl(l1 + 1, e) : nullptr;
    if (l2 && l2 + 1 < e && l2[1] != '+') throw Error(DRPRG_EAGAIN_SERIAL, "multi-line FASTQ");
}

// ---- gzip input ----------------------------------------------------------------------------------------------------
// libdeflate (whole-buffer inflate, 2-3x zlib's rate) is bound at run time: the image ships libdeflate.so.0 without its
// header; a host without it falls back to zlib's streaming inflate.
struct LibDeflate {
    void* (*alloc)() = nullptr;
    int (*gzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr; // 0 = success, 3 = output too small
    void (*release)(void*) = nullptr;
    bool ok() const { return alloc && gzip_ex && release; }
    static const LibDeflate& get()
    {
        static const LibDeflate inst = [] {
            LibDeflate l;
            if (std::getenv("DRPRG_HIP_NO_LIBDEFLATE")) return l; // (tests: force the zlib path)
            void* h = nullptr;
            for (const char* name : { "libdeflate.so.0", "libdeflate.so" })
                if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!h) return l;
            l.alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
            l.gzip_ex = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*)>(dlsym(h, "libdeflate_gzip_decompress_ex"));
            l.release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
            return l;
        }();
        return inst;
    }
};

// BGZF (bgzip, the usual way sequencing centres ship .fastq.gz): a series of gzip members of <= 64 KB, each with a 'BC' extra
// field that holds its compressed size -- the members can be located without inflating and inflated independently.
struct BgzfBlock {
    size_t in_off;
    uint32_t in_len, out_len;
    size_t out_off; // within its window
};

inline uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// size of the BGZF member that starts at p, 0 if p is not one
uint32_t bgzf_member_size(const unsigned char* p, size_t avail)
{
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const uint32_t xlen = (uint32_t)p[10] | ((uint32_t)p[11] << 8);
    if (12 + (size_t)xlen > avail) return 0;
    for (uint32_t o = 0; o + 4 <= xlen;) {
        const unsigned char* f = p + 12 + o;
        const uint32_t slen = (uint32_t)f[2] | ((uint32_t)f[3] << 8);
        if (f[0] == 'B' && f[1] == 'C' && slen == 2 && o + 6 <= xlen) return ((uint32_t)f[4] | ((uint32_t)f[5] << 8)) + 1;
        o += 4 + slen;
    }
    return 0;
}

// every member of the file, or an empty list if it is not BGZF from start to end
std::vector<BgzfBlock> bgzf_index(const unsigned char* data, size_t size)
{
    std::vector<BgzfBlock> blocks;
    size_t off = 0;
    while (off < size) {
        const uint32_t n = bgzf_member_size(data + off, size - off);
        if (n < 26 || off + n > size) return {};
        blocks.push_back(BgzfBlock { off, n, le32(data + off + n - 4), 0 });
        off += n;
    }
    return blocks;
}

void inflate_member(void* dec, const unsigned char* in, size_t in_len, char* out, size_t out_len, const std::string& path)
{
    size_t used = 0, got = 0;
    const int rc = LibDeflate::get().gzip_ex(dec, in, in_len, out, out_len, &used, &got);
    if (rc != 0 || got != out_len) throw Error(DRPRG_EIO, "corrupt gzip block in " + path);
}

// text buffers of the plain-file parser threads, kept between calls (a buffer is ~48 MB of huge pages; at most 24 are kept)
BufferPool& plain_pool()
{
    static BufferPool pool;
    return pool;
}

bool detect_format(const char* p, const char* e, bool& fastq)
{
    while (p < e && (*p == '\n' || *p == '\r' || *p == ' ')) ++p;
    if (p >= e) return false;
    if (*p == '@') fastq = true;
    else if (*p == '>') fastq = false;
    else throw Error(DRPRG_EFORMAT, "reads file is neither FASTA nor FASTQ");
    return true;
}

} // namespace

IngestStats ingest_fastx(const std::string& path, int threads, const IngestHooks& hooks)
{
    if (threads < 1) threads = 1;
    Shared sh(hooks);
    BufferPool gz_pool; // (outlives the queue and the threads that hand window buffers back to it)
    IngestStats st;
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(DRPRG_ENOENT, "cannot open reads file " + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        throw Error(DRPRG_EIO, "cannot stat " + path);
    }
    unsigned char magic[2] = { 0, 0 };
    const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    // (gzip: at most 8 parser threads; every queued window is a 32 MB buffer that is recycled only once parsed, so a short queue keeps the same few buffers -- and their pages -- going round)
    SliceQueue queue(gz ? (size_t)std::min(threads, 8) + 2 : (size_t)threads + 2);
    bool fastq = true;
    std::atomic<bool> format_known { false };

    auto worker = [&]() {
        Block blk;
        std::shared_ptr<TextBuffer> own_text;
        try {
            Slice s;
            while (queue.pop(s)) {
                if (sh.failed) continue; // drain
                if (!blk.bases) {
                    const int64_t t0 = sh.debug ? sh.now_ns() : 0;
                    blk = sh.new_block();
                    if (sh.debug) sh.ns_alloc += sh.now_ns() - t0;
                }
                const int64_t t0 = sh.debug ? sh.now_ns() : 0;
                if (!s.begin && s.file_len) { // plain file: this thread reads the slice into its own (recycled, huge-page) buffer
                    if (!own_text || own_text->size() < s.file_len + 64) own_text = plain_pool().acquire(std::max<size_t>(s.file_len + 64, plain_slice_bytes() + (plain_slice_bytes() >> 1)));
                    size_t have = 0;
                    while (have < s.file_len) {
                        const ssize_t r = pread(fd, own_text->data() + have, s.file_len - have, (off_t)(s.file_off + have));
                        if (r <= 0) throw Error(DRPRG_EIO, "cannot read " + path);
                        have += (size_t)r;
                    }
                    if (sh.debug) sh.ns_read += sh.now_ns() - t0;
                    parse_slice(own_text->data(), own_text->data() + s.file_len, fastq, blk, sh);
                } else parse_slice(s.begin, s.end, fastq, blk, sh);
                s = Slice(); // (a mapped slice: its mapping goes now, on this thread)
                if (sh.debug) sh.ns_parse += sh.now_ns() - t0;
            }
            if (!sh.failed) sh.submit(blk);
        } catch (const Error& e) {
            sh.fail(e.code, e.what());
            queue.close();
        } catch (const std::exception& e) {
            sh.fail(DRPRG_EIO, e.what());
            queue.close();
        }
        sh.free_block(blk);
    };

    void* map = nullptr;
    size_t map_len = 0;
    std::vector<std::thread> pool;
    try {
        if (!gz) {
            map_len = (size_t)sb.st_size;
            if (map_len == 0) {
                close(fd);
                return st;
            }
            // The file is mapped slice by slice, and every slice's mapping is torn down by the parser thread that read it: one
            // mapping of the whole file costs ~25 ns per 4 KB page to unmap -- 10 M x 150 bp of FASTQ text are 3.2 GB, 770 k
            // pages, 40-75 ms of munmap on the calling thread after the last read was parsed (measured: the parser threads were
            // busy for 25 of a call's 90 ms).  A slice's mapping starts at the page that holds the slice's first byte and reaches
            // a quarter slice past its nominal end, where the next record start is looked for.
            const size_t page = (size_t)sysconf(_SC_PAGESIZE);
            {
                // format and layout from the head of the file (8 MB at a time until something other than blank lines shows up)
                bool known = false;
                for (size_t at = 0; at < map_len && !known; at += (size_t)8 << 20) {
                    const size_t n = std::min(map_len - at, (size_t)8 << 20);
                    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, (off_t)at);
                    if (m == MAP_FAILED) throw Error(DRPRG_EIO, "cannot mmap " + path);
                    struct Unmap {
                        void* p;
                        size_t n;
                        ~Unmap() { munmap(p, n); }
                    } unmap { m, n };
                    const char* text = (const char*)m;
                    if (detect_format(text, text + n, fastq)) {
                        known = true;
                        if (fastq) require_four_line_fastq(text, text + n);
                    }
                }
                if (!known) {
                    close(fd);
                    return st;
                }
            }
            // (a worker pins its own block -- a few milliseconds of driver time, serialised -- so every worker should have several
            // slices to fill it with: measured on 4 M reads, 8 workers 50 ms, 32 workers 139 ms)
            const int n_workers = (int)std::min<size_t>((size_t)threads, map_len / (3 * SLICE_BYTES) + 1);
            for (int t = 0; t < n_workers; ++t) pool.emplace_back(worker);
            size_t cur = 0; // file offset of the next slice's first byte (a record start)
            // Default: no mapping at all -- the parser thread reads its slice with pread into a buffer it keeps (no page-table
            // set-up and tear-down per 4 KB page of the file, which is what a mapping costs however it is cut: 40-50 ms per thread
            // with 32 threads under one address-space lock against ~15 ms for the copy); the cut points come from small reads
            // around the nominal slice ends.  DRPRG_INGEST_MMAP=1: the per-slice mappings below.
            static const bool use_mmap = [] {
                const char* e = std::getenv("DRPRG_INGEST_MMAP");
                return e && std::atoi(e) != 0;
            }();
            std::vector<char> win;
            while (!use_mmap && cur < map_len && !sh.failed) {
                size_t cut = map_len;
                const size_t slice = plain_slice_bytes();
                if (map_len - cur > slice + (slice >> 2)) {
                    for (size_t look = (size_t)1 << 18;; look *= 4) { // the next record start at or after cur + slice
                        const size_t at = cur + slice, n = std::min(look, map_len - at);
                        win.resize(n);
                        size_t have = 0;
                        while (have < n) {
                            const ssize_t r = pread(fd, win.data() + have, n - have, (off_t)(at + have));
                            if (r <= 0) throw Error(DRPRG_EIO, "cannot read " + path);
                            have += (size_t)r;
                        }
                        const char* c = next_record(win.data(), win.data() + n, fastq);
                        if (c < win.data() + n) {
                            cut = at + (size_t)(c - win.data());
                            break;
                        }
                        if (at + n >= map_len) { // the rest of the file holds no further record start
                            cut = map_len;
                            break;
                        }
                        if (look > ((size_t)1 << 30)) throw Error(DRPRG_EFORMAT, "no record boundary in 1 GB of " + path);
                    }
                }
                Slice sl;
                sl.file_off = cur;
                sl.file_len = cut - cur;
                cur = cut;
                queue.push(std::move(sl));
            }
            while (cur < map_len && !sh.failed) {
                size_t want = SLICE_BYTES + (SLICE_BYTES >> 2);
                for (;;) {
                    const size_t lo = cur / page * page;
                    const size_t hi = std::min(map_len, cur + want);
                    void* m = mmap(nullptr, hi - lo, PROT_READ, MAP_PRIVATE, fd, (off_t)lo);
                    if (m == MAP_FAILED) throw Error(DRPRG_EIO, "cannot mmap " + path);
                    const size_t m_len = hi - lo;
                    std::shared_ptr<void> owner(m, [m_len](void* q) { munmap(q, m_len); });
                    const char* b = (const char*)m + (cur - lo);
                    const char* e = (const char*)m + m_len;
                    const char* cut = e;
                    if (hi < map_len) {
                        cut = next_record(b + std::min(SLICE_BYTES, (size_t)(e - b) - 1), e, fastq);
                        if (cut >= e) { // no record start in the last quarter (a very long read): map more and look again
                            want *= 2;
                            continue;
                        }
                    }
                    Slice sl;
                    sl.begin = b;
                    sl.end = cut;
                    sl.mapping = std::move(owner);
                    cur += (size_t)(cut - b);
                    queue.push(std::move(sl));
                    break;
                }
            }
        } else {
            // Four ways to inflate: (1) BGZF: the members are located from their headers and inflated in parallel, a window
            // of text at a time; (4) a plain gzip file of some size with several threads to spend: all threads inflate the one
            // stream, entering it at block boundaries found in the compressed data (pgunzip.h); (2) a small plain gzip member
            // whose size field can be trusted: one libdeflate call into one buffer; (3) anything else, or no libdeflate on
            // this host: zlib's streaming inflate on this thread.
            const LibDeflate& ld = LibDeflate::get();
            const size_t gz_len = (size_t)sb.st_size;
            const char* par_env = std::getenv("DRPRG_GZ_PARALLEL"); // 0: never way 4; 1: also for small files (tests)
            const bool par_forced = par_env && std::atoi(par_env) > 0;
            const bool par_ok = !(par_env && std::atoi(par_env) == 0) && threads >= 2 && (par_forced || gz_len >= (size_t(4) << 20));
            void* gz_map = (ld.ok() || par_ok) && gz_len >= 18 ? mmap(nullptr, gz_len, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
            const unsigned char* gz_data = gz_map != MAP_FAILED ? (const unsigned char*)gz_map : nullptr;
            struct Unmap {
                void* p;
                size_t n;
                ~Unmap() { if (p && p != MAP_FAILED) munmap(p, n); }
            } unmap { gz_map, gz_len };
            std::vector<BgzfBlock> blocks;
            if (gz_data && ld.ok()) blocks = bgzf_index(gz_data, gz_len);
            std::unique_ptr<ParallelGunzip> pgz; // way (4)
            if (gz_data && blocks.empty() && par_ok) {
                const char* chunk_env = std::getenv("DRPRG_GZ_CHUNK");
                pgz.reset(new ParallelGunzip(gz_data, gz_len, threads, chunk_env ? (size_t)std::atoll(chunk_env) : 0));
            }
            std::shared_ptr<std::vector<char>> whole; // way (2)
            if (gz_data && ld.ok() && blocks.empty() && !pgz && gz_len < (size_t(1) << 30)) {
                const size_t isize = le32(gz_data + gz_len - 4);
                if (isize >= gz_len / 2) { // (a wrapped size field of a > 4 GB stream is most likely smaller than that)
                    auto buf = std::make_shared<std::vector<char>>(isize + 1);
                    void* dec = ld.alloc();
                    size_t used = 0, got = 0;
                    const int rc = dec ? ld.gzip_ex(dec, gz_data, gz_len, buf->data(), isize, &used, &got) : 1;
                    if (dec) ld.release(dec);
                    if (rc == 0 && used == gz_len && got == isize) {
                        buf->resize(isize);
                        whole = buf;
                    } // (several members, or a wrapped size: the streaming reader takes it)
                }
            }
            for (int t = 0; t < std::min(threads, 8); ++t) pool.emplace_back(worker); // (the inflater feeds them: more only pin more blocks)
            // text source: fills dst with up to cap bytes of inflated text, returns the number written (0 = end)
            gzFile gzf = nullptr;
            std::unique_ptr<Crew> bgzf_crew;
            size_t next_block = 0, whole_off = 0;
            std::function<size_t(char*, size_t)> read_text;
            if (!blocks.empty()) {
                st.gz_mode = 1;
                read_text = [&](char* dst, size_t cap) -> size_t {
                    size_t first = next_block, total = 0;
                    while (next_block < blocks.size() && total + blocks[next_block].out_len <= cap) {
                        blocks[next_block].out_off = total;
                        total += blocks[next_block].out_len;
                        ++next_block;
                    }
                    if (next_block == first && next_block < blocks.size()) throw Error(DRPRG_EIO, "BGZF block larger than a slice in " + path);
                    const size_t last = next_block;
                    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, (last - first + 15) / 16)); // >= 16 members (1 MB of text) per thread
                    std::atomic<size_t> cursor { first };
                    std::atomic<bool> bad { false };
                    auto job = [&]() {
                        thread_local struct Dec { // (one decompressor per thread of the crew, not one per window)
                            const LibDeflate* l = nullptr;
                            void* d = nullptr;
                            ~Dec() { if (d && l) l->release(d); }
                        } dec;
                        if (!dec.d) {
                            dec.l = &ld;
                            dec.d = ld.alloc();
                        }
                        if (!dec.d) { bad = true; return; }
                        try {
                            for (size_t b; (b = cursor.fetch_add(2)) < last;) // (two members = 128 KB of text per grab: the window ends on a barrier)
                                for (size_t i = b; i < std::min(b + 2, last); ++i)
                                    inflate_member(dec.d, gz_data + blocks[i].in_off, blocks[i].in_len, dst + blocks[i].out_off, blocks[i].out_len, path);
                        } catch (const Error&) {
                            bad = true;
                        }
                    };
                    if (!bgzf_crew) bgzf_crew.reset(new Crew(threads - 1)); // (threads that stay: 39 windows x 31 thread starts otherwise)
                    bgzf_crew->run(job, nt);
                    if (bad) throw Error(DRPRG_EIO, "corrupt gzip block in " + path);
                    return total;
                };
            } else if (pgz) {
                st.gz_mode = 4;
                read_text = [&](char* dst, size_t cap) -> size_t {
                    size_t have = 0;
                    while (have < cap) { // (a call hands over what one round of chunks left; a short window would read as the end)
                        const size_t n = pgz->read(dst + have, cap - have);
                        if (n == 0) break;
                        have += n;
                    }
                    return have;
                };
            } else if (whole) {
                st.gz_mode = 2;
                read_text = [&](char* dst, size_t cap) -> size_t {
                    const size_t n = std::min(cap, whole->size() - whole_off);
                    std::memcpy(dst, whole->data() + whole_off, n);
                    whole_off += n;
                    return n;
                };
            } else {
                st.gz_mode = 3;
                gzf = gzdopen(dup(fd), "rb");
                if (!gzf) throw Error(DRPRG_EIO, "cannot read gzip stream " + path);
                gzbuffer(gzf, 1 << 20);
                read_text = [&](char* dst, size_t cap) -> size_t {
                    size_t have = 0;
                    while (have < cap) {
                        const int n = gzread(gzf, dst + have, (unsigned)std::min<size_t>(cap - have, 1u << 30));
                        if (n < 0) throw Error(DRPRG_EIO, "gzip read error in " + path);
                        if (n == 0) break;
                        have += (size_t)n;
                    }
                    return have;
                };
            }
            struct CloseGz {
                gzFile& f;
                ~CloseGz() { if (f) gzclose(f); }
            } close_gz { gzf };
            std::vector<char> carry;
            bool eof = false;
            int64_t ns_text = 0, ns_acquire = 0, ns_push = 0, t_mark = 0; // (DRPRG_INGEST_DEBUG: where the window loop's time goes)
            struct LoopTimes {
                Shared& sh;
                int64_t &text, &acquire, &push;
                ~LoopTimes()
                {
                    if (sh.debug)
                        std::fprintf(stderr, "[ingest] compressed input, window loop of the calling thread until %.1f ms: inflating %.1f ms, window buffers %.1f ms, "
                                             "handing windows to the parser threads %.1f ms\n", sh.now_ns() / 1e6, text / 1e6, acquire / 1e6, push / 1e6);
                }
            } loop_times { sh, ns_text, ns_acquire, ns_push };
            while (!eof && !sh.failed) {
                // (one size for every window -- the carry is at most the 1 MB the cut is looked for in --, so a recycled buffer always fits)
                t_mark = sh.debug ? sh.now_ns() : 0;
                auto buf = gz_pool.acquire(std::max(carry.size(), size_t(1) << 20) + SLICE_BYTES);
                if (!carry.empty()) std::memcpy(buf->data(), carry.data(), carry.size());
                size_t have = carry.size();
                carry.clear();
                if (sh.debug) ns_acquire += sh.now_ns() - t_mark;
                t_mark = sh.debug ? sh.now_ns() : 0;
                const size_t got = read_text(buf->data() + have, SLICE_BYTES);
                if (sh.debug) ns_text += sh.now_ns() - t_mark;
                have += got;
                if (got < SLICE_BYTES - (1u << 17)) // a short window: the source has nothing left (BGZF windows end up to 64 KB short)
                    eof = blocks.empty() ? true : next_block >= blocks.size();
                if (have == 0) break;
                const char* b = buf->data();
                const char* e = b + have;
                if (!format_known) {
                    if (!detect_format(b, e, fastq)) continue;
                    if (fastq) require_four_line_fastq(b, e);
                    format_known = true;
                }
                const char* cut = e;
                if (!eof) {
                    // last record start in the final MB of the buffer: everything after it is carried over
                    const char* scan = have > (1u << 20) ? e - (1u << 20) : b;
                    const char* last = nullptr;
                    for (const char* p = next_record(scan, e, fastq); p < e; p = next_record(p, e, fastq)) last = p;
                    if (!last) throw Error(DRPRG_EFORMAT, "no record boundary in 1 MB of " + path + " (record too long)");
                    cut = last;
                    carry.assign(cut, e);
                }
                t_mark = sh.debug ? sh.now_ns() : 0;
                queue.push(Slice { b, cut, buf });
                if (sh.debug) ns_push += sh.now_ns() - t_mark;
            }
        }
    } catch (const Error& e) {
        sh.fail(e.code, e.what());
    }
    const int64_t t_loop_end = sh.debug ? sh.now_ns() : 0;
    queue.close();
    for (auto& t : pool) t.join();
    if (sh.debug) std::fprintf(stderr, "[ingest] parser threads done %.1f ms after the last slice was queued (at %.1f ms)\n", (sh.now_ns() - t_loop_end) / 1e6, t_loop_end / 1e6);
    if (map && map != MAP_FAILED) munmap(map, map_len);
    close(fd);
    if (sh.failed) throw Error(sh.error_code, sh.error);
    st.reads = sh.reads;
    st.bases = sh.bases;
    st.batches = sh.batches;
    st.parallel = threads > 1;
    if (sh.debug)
        std::fprintf(stderr, "[ingest] wall %.1f ms, %llu batches, first hand-over at %.1f ms; summed over the %zu parser threads: parse (hand-overs included) %.1f ms, "
                             "waiting for the submitter %.1f ms, inside the submitter %.1f ms, block allocation %.1f ms, reading the file %.1f ms\n", sh.now_ns() / 1e6,
            (unsigned long long)st.batches, sh.ns_first_submit.load() / 1e6, pool.size(), sh.ns_parse.load() / 1e6, sh.ns_submit_wait.load() / 1e6,
            sh.ns_submit_run.load() / 1e6, sh.ns_alloc.load() / 1e6, sh.ns_read.load() / 1e6);
    return st;
}

} // namespace drprg
