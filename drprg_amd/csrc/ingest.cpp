// ingest.cpp -- see ingest.h
#include "ingest.h"
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <memory>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>

namespace drprg {

namespace {

// A slice of FASTQ text is about half bases, so one slice normally fills one block (one H2D copy + one launch
// sequence per slice).  Pinning memory is slow (~10 GB/s), hence small blocks, allocated only by workers that get work.
constexpr size_t SLICE_BYTES = 32u << 20; // text handed to one parser task
constexpr size_t BLOCK_BASES = 24u << 20; // pinned block: bases capacity ...
constexpr size_t BLOCK_READS = 1u << 20;  // ... and read capacity (flushed early when either fills up)

struct Block {
    uint8_t* bases = nullptr;
    uint64_t* offsets = nullptr;
    uint64_t n_reads = 0, n_bases = 0;
};

struct Shared {
    const IngestHooks& hooks;
    std::mutex submit_mu, err_mu;
    std::atomic<uint64_t> reads { 0 }, bases { 0 }, batches { 0 };
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
        {
            std::lock_guard<std::mutex> g(submit_mu);
            hooks.submit(pb);
        }
        reads += b.n_reads;
        bases += b.n_bases;
        batches += 1;
        b.n_reads = 0;
        b.n_bases = 0;
    }
    Block new_block()
    {
        Block b;
        const size_t bytes_b = BLOCK_BASES + 64, bytes_o = (BLOCK_READS + 2) * sizeof(uint64_t);
        b.bases = (uint8_t*)(hooks.alloc ? hooks.alloc(bytes_b) : std::malloc(bytes_b));
        b.offsets = (uint64_t*)(hooks.alloc ? hooks.alloc(bytes_o) : std::malloc(bytes_o));
        if (!b.bases || !b.offsets) throw Error(DRPRG_ENOMEM, "cannot allocate an ingest block");
        b.offsets[0] = 0;
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

// parse the records of [p, e) (whole records only) into the block, submitting whenever it fills up
void parse_slice(const char* p, const char* e, bool fastq, Block& blk, Shared& sh)
{
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
            if (blk.n_bases + len > BLOCK_BASES || blk.n_reads + 1 > BLOCK_READS) sh.submit(blk);
            std::memcpy(blk.bases + blk.n_bases, cursor, len);
            blk.n_bases += len;
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
            if (blk.n_bases + bound > BLOCK_BASES || blk.n_reads + 1 > BLOCK_READS) sh.submit(blk);
            while (cursor < rec_end) {
                const char* nl = find_nl(cursor, rec_end);
                const char* stop = nl ? nl : rec_end;
                const char* next = nl ? nl + 1 : rec_end;
                strip_cr(cursor, stop);
                std::memcpy(blk.bases + blk.n_bases, cursor, (size_t)(stop - cursor));
                blk.n_bases += (size_t)(stop - cursor);
                cursor = next;
            }
            blk.offsets[++blk.n_reads] = blk.n_bases;
            p = rec_end;
        }
    }
}

struct Slice {
    const char* begin = nullptr;
    const char* end = nullptr;
    std::shared_ptr<std::vector<char>> owner; // inflated gzip text; null for a memory-mapped file
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

// multi-line FASTQ (sequence wrapped over several lines) is left to the serial reader
void require_four_line_fastq(const char* p, const char* e)
{
    while (p < e && (*p == '\n' || *p == '\r' || *p == ' ')) ++p;
    const char* l1 = find_nl(p, e);
    const char* l2 = l1 ? find_nl(l1 + 1, e) : nullptr;
    if (l2 && l2 + 1 < e && l2[1] != '+') throw Error(DRPRG_EAGAIN_SERIAL, "multi-line FASTQ");
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
    SliceQueue queue((size_t)threads + 2);
    bool fastq = true;
    std::atomic<bool> format_known { false };

    auto worker = [&]() {
        Block blk;
        try {
            Slice s;
            while (queue.pop(s)) {
                if (sh.failed) continue; // drain
                if (!blk.bases) blk = sh.new_block();
                parse_slice(s.begin, s.end, fastq, blk, sh);
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
            map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) throw Error(DRPRG_EIO, "cannot mmap " + path);
            madvise(map, map_len, MADV_SEQUENTIAL);
            const char* text = (const char*)map;
            const char* end = text + map_len;
            if (!detect_format(text, end, fastq)) {
                munmap(map, map_len);
                close(fd);
                return st;
            }
            if (fastq) require_four_line_fastq(text, end);
            const int n_workers = (int)std::min<size_t>((size_t)threads, map_len / SLICE_BYTES + 1);
            for (int t = 0; t < n_workers; ++t) pool.emplace_back(worker);
            const char* cur = text;
            while (cur < end && !sh.failed) {
                const char* cut = end;
                if ((size_t)(end - cur) > SLICE_BYTES + (SLICE_BYTES >> 2)) cut = next_record(cur + SLICE_BYTES, end, fastq);
                queue.push(Slice { cur, cut, nullptr });
                cur = cut;
            }
        } else {
            gzFile gzf = gzdopen(dup(fd), "rb");
            if (!gzf) throw Error(DRPRG_EIO, "cannot read gzip stream " + path);
            gzbuffer(gzf, 1 << 20);
            for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
            std::vector<char> carry;
            bool eof = false;
            while (!eof && !sh.failed) {
                auto buf = std::make_shared<std::vector<char>>();
                buf->resize(carry.size() + SLICE_BYTES);
                if (!carry.empty()) std::memcpy(buf->data(), carry.data(), carry.size());
                size_t have = carry.size();
                carry.clear();
                while (have < buf->size()) {
                    int n = gzread(gzf, buf->data() + have, (unsigned)std::min<size_t>(buf->size() - have, 1u << 30));
                    if (n < 0) {
                        gzclose(gzf);
                        throw Error(DRPRG_EIO, "gzip read error in " + path);
                    }
                    if (n == 0) {
                        eof = true;
                        break;
                    }
                    have += (size_t)n;
                }
                buf->resize(have);
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
                queue.push(Slice { b, cut, buf });
            }
            gzclose(gzf);
        }
    } catch (const Error& e) {
        sh.fail(e.code, e.what());
    }
    queue.close();
    for (auto& t : pool) t.join();
    if (map && map != MAP_FAILED) munmap(map, map_len);
    close(fd);
    if (sh.failed) throw Error(sh.error_code, sh.error);
    st.reads = sh.reads;
    st.bases = sh.bases;
    st.batches = sh.batches;
    st.parallel = threads > 1;
    return st;
}

} // namespace drprg
