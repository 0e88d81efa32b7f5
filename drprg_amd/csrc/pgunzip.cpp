// pgunzip.cpp -- see pgunzip.h.  RFC 1951 / RFC 1952 decoder written for this purpose: table-driven Huffman decoding into
// 16-bit symbols (bytes and window markers), a block-start finder, the stitcher and the resolver.
#include "pgunzip.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <dlfcn.h>
#include <sys/mman.h>
#include <emmintrin.h>
#include <thread>
#include <vector>
#include <zlib.h>

namespace drprg {

namespace {

constexpr unsigned LIT_BITS = 11, DIST_BITS = 8;
constexpr size_t WINDOW = 32768;
constexpr uint16_t MARK = 0x8000;                 // symbol MARK + i = byte i of the 32 KB before the segment
constexpr size_t LIT_TABLE = (1u << LIT_BITS) + 288 * 16, DIST_TABLE = (1u << DIST_BITS) + 32 * 128;
constexpr size_t HARD_CAP = size_t(1) << 31;      // symbols in one segment (a block that long is not a FASTQ file)

enum Kind : uint8_t { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_LINK = 3, K_BAD = 4 };
struct Entry {
    uint16_t val;  // literal byte / base length / base distance / offset of the second-level table
    uint8_t bits;  // length of the whole code
    uint8_t op;    // kind | (extra bits, or bits of the second-level table) << 4
};
inline unsigned kind(Entry e) { return e.op & 15u; }
inline unsigned extra(Entry e) { return e.op >> 4; }

const uint16_t LEN_BASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t DIST_BASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
const uint8_t CL_ORDER[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

inline uint32_t bit_reverse(uint32_t code, unsigned len)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman code of lens[0..n) as a two-level table indexed by the next bits of the stream (LSB first).
// Accepts what zlib accepts: a complete code, one single code of length 1, or (allow_empty) no code at all.
template <class MakeEntry>
bool build_table(const uint8_t* lens, unsigned n, unsigned primary, Entry* table, size_t table_cap, bool allow_empty, MakeEntry make)
{
    unsigned count[16] = { 0 };
    for (unsigned i = 0; i < n; ++i) ++count[lens[i]];
    unsigned max_len = 15;
    while (max_len > 0 && count[max_len] == 0) --max_len;
    const Entry bad { 0, 1, K_BAD };
    const size_t psize = size_t(1) << primary;
    if (max_len == 0) {
        if (!allow_empty) return false;
        std::fill(table, table + psize, bad);
        return true;
    }
    int left = 1;
    for (unsigned l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)count[l];
        if (left < 0) return false; // over-subscribed
    }
    if (left > 0 && max_len != 1) return false; // incomplete
    uint32_t next[16];
    uint32_t code = 0;
    for (unsigned l = 1; l <= 15; ++l) {
        code = (code + count[l - 1] * (l > 1)) << 1;
        next[l] = code;
    }
    std::fill(table, table + psize, bad);
    size_t used = psize;
    if (max_len > primary) { // second-level tables: as wide as the longest code that shares the first `primary` bits
        uint8_t sub[1u << LIT_BITS] = { 0 };
        uint32_t nx[16];
        std::memcpy(nx, next, sizeof nx);
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            const uint32_t r = bit_reverse(nx[l]++, l);
            if (l > primary) sub[r & (psize - 1)] = std::max<uint8_t>(sub[r & (psize - 1)], (uint8_t)(l - primary));
        }
        for (size_t p = 0; p < psize; ++p)
            if (sub[p]) {
                const size_t sz = size_t(1) << sub[p];
                if (used + sz > table_cap) return false;
                table[p] = Entry { (uint16_t)used, (uint8_t)primary, (uint8_t)(K_LINK | (sub[p] << 4)) };
                std::fill(table + used, table + used + sz, bad);
                used += sz;
            }
    }
    for (unsigned s = 0; s < n; ++s) {
        const unsigned l = lens[s];
        if (!l) continue;
        const uint32_t r = bit_reverse(next[l]++, l);
        Entry e = make(s);
        e.bits = (uint8_t)l;
        if (l <= primary) {
            for (size_t i = r; i < psize; i += size_t(1) << l) table[i] = e;
        } else {
            const Entry link = table[r & (psize - 1)];
            const unsigned sb = extra(link), rest = l - primary;
            for (size_t i = r >> primary; i < (size_t(1) << sb); i += size_t(1) << rest) table[link.val + i] = e;
        }
    }
    return true;
}

inline Entry lit_entry(unsigned s)
{
    if (s < 256) return Entry { (uint16_t)s, 0, K_LIT };
    if (s == 256) return Entry { 0, 0, K_EOB };
    if (s < 286) return Entry { LEN_BASE[s - 257], 0, (uint8_t)(K_BASE | (LEN_EXTRA[s - 257] << 4)) };
    return Entry { 0, 0, K_BAD };
}
inline Entry dist_entry(unsigned s)
{
    if (s < 30) return Entry { DIST_BASE[s], 0, (uint8_t)(K_BASE | (DIST_EXTRA[s] << 4)) };
    return Entry { 0, 0, K_BAD };
}

struct Tables {
    Entry lit[LIT_TABLE];
    Entry dist[DIST_TABLE];
};

const Tables& fixed_tables()
{
    static const Tables* t = [] {
        Tables* f = new Tables;
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(l, 288, LIT_BITS, f->lit, LIT_TABLE, false, lit_entry);
        build_table(d, 32, DIST_BITS, f->dist, DIST_TABLE, false, dist_entry);
        return f;
    }();
    return *t;
}

// LSB-first bit reader over the mapped file.  Past the end it supplies zero bits; callers compare pos() with the length.
struct BitReader {
    const uint8_t* base;
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    unsigned n = 0;
    uint64_t past = 0; // virtual zero bytes supplied after the end
    BitReader(const uint8_t* b, size_t len) : base(b), p(b), end(b + len) {}
    void seek(uint64_t bit)
    {
        const uint64_t byte = bit >> 3, size = (uint64_t)(end - base);
        p = base + std::min(byte, size);
        past = byte > size ? byte - size : 0;
        buf = 0;
        n = 0;
        refill();
        consume((unsigned)(bit & 7));
    }
    inline void refill()
    {
        if (end - p >= 8) {
            uint64_t v;
            std::memcpy(&v, p, 8);
            buf |= v << n;
            p += (63 - n) >> 3;
            n |= 56;
        } else {
            while (n <= 56) {
                if (p < end) buf |= (uint64_t)*p++ << n;
                else ++past; // (counted as read: pos() then exceeds the length)
                n += 8;
            }
        }
    }
    inline void consume(unsigned k)
    {
        buf >>= k;
        n -= k;
    }
    inline uint32_t take(unsigned k) // k <= 32, after a refill
    {
        const uint32_t v = (uint32_t)(buf & ((uint64_t(1) << k) - 1));
        consume(k);
        return v;
    }
    uint64_t pos() const { return ((uint64_t)(p - base) + past) * 8 - n; }
    void align_byte() { consume(n & 7); }
};

// code lengths of a dynamic block (the header after BFINAL / BTYPE); false = not a valid header
bool read_dynamic_lengths(BitReader& br, uint8_t* lens /* [320] */, unsigned& hlit, unsigned& hdist)
{
    br.refill();
    hlit = br.take(5) + 257;
    hdist = br.take(5) + 1;
    const unsigned hclen = br.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = { 0 };
    for (unsigned i = 0; i < hclen; ++i) {
        if (br.n < 3) br.refill();
        cl[CL_ORDER[i]] = (uint8_t)br.take(3);
    }
    Entry clt[128];
    {
        // (zlib: the code-length code must be complete)
        int left = 1;
        unsigned count[8] = { 0 };
        for (unsigned i = 0; i < 19; ++i) ++count[cl[i]];
        for (unsigned l = 1; l <= 7; ++l) {
            left = (left << 1) - (int)count[l];
            if (left < 0) return false;
        }
        if (left != 0) return false;
    }
    if (!build_table(cl, 19, 7, clt, 128, false, [](unsigned s) { return Entry { (uint16_t)s, 0, K_LIT }; })) return false;
    const unsigned total = hlit + hdist;
    unsigned i = 0;
    while (i < total) {
        br.refill();
        const Entry e = clt[br.buf & 127];
        if (kind(e) != K_LIT) return false;
        br.consume(e.bits);
        const unsigned s = e.val;
        if (s < 16) {
            lens[i++] = (uint8_t)s;
            continue;
        }
        unsigned rep, v = 0;
        if (s == 16) {
            if (i == 0) return false;
            v = lens[i - 1];
            rep = 3 + br.take(2);
        } else if (s == 17) rep = 3 + br.take(3);
        else rep = 11 + br.take(7);
        if (i + rep > total) return false;
        while (rep--) lens[i++] = (uint8_t)v;
    }
    return lens[256] != 0; // a block without an end-of-block code never ends
}

// Symbol buffers are tens of MB written once front to back: 2 MB-aligned and advised as huge pages, so that filling them costs
// one page fault per 2 MB instead of 512 (with every thread of a round faulting at once the kernel serialises them).
constexpr size_t HUGE = size_t(2) << 20;
uint16_t* alloc_symbols(size_t n)
{
    const size_t bytes = (n * sizeof(uint16_t) + HUGE - 1) / HUGE * HUGE;
    void* p = std::aligned_alloc(HUGE, bytes);
    if (p) madvise(p, bytes, MADV_HUGEPAGE);
    return (uint16_t*)p;
}

struct MemberEnd {
    size_t at;      // symbols of the segment that belong to members ended so far
    uint32_t crc, isize;
};

struct Segment {
    uint64_t start_bit = 0, end_bit = 0; // both block boundaries
    uint16_t* buf = nullptr;             // [WINDOW marker symbols][n symbols]
    size_t cap = 0, n = 0;
    bool ok = false, eof = false, hit_cap = false;
    std::vector<MemberEnd> ends;
    std::vector<uint8_t> window;         // the 32 KB before symbol 0, known once the predecessors are resolved
    size_t read_pos = 0, ends_done = 0;  // resolver cursor: symbols handed out, member ends checked
    std::string error;
    ~Segment() { std::free(buf); }
    Segment() = default;
    Segment(uint16_t* b, size_t c) : buf(b), cap(c) {}
    Segment(const Segment&) = delete;
    Segment& operator=(const Segment&) = delete;
};

// Symbol buffers outlive a ParallelGunzip: handing ~1 GB of them back to the kernel took 25 ms at the end of every file, and the
// next file faulted the same pages in again.  The process keeps at most CACHE_BYTES of them (the largest first to go).
struct SymbolCache {
    static constexpr size_t CACHE_BYTES = size_t(3) << 29; // 1.5 GB
    std::mutex mu;
    std::vector<std::pair<uint16_t*, size_t>> bufs; // (pointer, capacity in symbols)
    size_t bytes = 0;
    static SymbolCache& get()
    {
        static SymbolCache* c = new SymbolCache; // (never destroyed: nothing to gain from unmapping at exit)
        return *c;
    }
    bool take(std::pair<uint16_t*, size_t>& out)
    {
        std::lock_guard<std::mutex> g(mu);
        if (bufs.empty()) return false;
        out = bufs.back();
        bufs.pop_back();
        bytes -= out.second * sizeof(uint16_t);
        return true;
    }
    void give(uint16_t* p, size_t cap)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            if (bytes + cap * sizeof(uint16_t) <= CACHE_BYTES) {
                bufs.push_back({ p, cap });
                bytes += cap * sizeof(uint16_t);
                return;
            }
        }
        std::free(p);
    }
};

enum class Status { OK, BAD_FIRST_BLOCK, FAILED };

// RFC 1952 member header at byte offset `off`; returns the offset of the deflate data, 0 if there is no valid header
size_t gzip_header(const uint8_t* d, size_t len, size_t off)
{
    if (off + 18 > len || d[off] != 0x1f || d[off + 1] != 0x8b || d[off + 2] != 8) return 0;
    const unsigned flg = d[off + 3];
    if (flg & 0xE0) return 0;
    size_t p = off + 10;
    if (flg & 4) {
        if (p + 2 > len) return 0;
        p += 2 + ((size_t)d[p] | ((size_t)d[p + 1] << 8));
    }
    for (unsigned bit : { 8u, 16u })
        if (flg & bit) {
            while (p < len && d[p]) ++p;
            ++p;
        }
    if (flg & 2) p += 2;
    return p < len ? p : 0;
}

struct Inflater {
    const uint8_t* data;
    size_t len;
    Tables dyn; // (per thread: 40 KB)

    bool grow(Segment& s, size_t need_free)
    {
        const size_t want = WINDOW + s.n + need_free;
        if (want <= s.cap) return true;
        size_t cap = std::max<size_t>(s.cap * 2, want + (want >> 1));
        if (cap > HARD_CAP + WINDOW + (1u << 20)) cap = HARD_CAP + WINDOW + (1u << 20);
        if (cap < want) return false;
        uint16_t* nb = alloc_symbols(cap);
        if (!nb) return false;
        std::memcpy(nb, s.buf, (WINDOW + s.n) * sizeof(uint16_t));
        std::free(s.buf);
        s.buf = nb;
        s.cap = cap;
        return true;
    }

    // Inflates from the block boundary start_bit (at_header: a member header at that byte instead) up to the first
    // dynamic, non-final block header at or after stop_bit, the end of the stream, or the first block boundary after
    // soft_cap symbols.  first_block_probe: an error inside the first block is reported as BAD_FIRST_BLOCK.
    Status run(Segment& s, uint64_t start_bit, bool at_header, uint64_t stop_bit, size_t soft_cap, bool first_block_probe, size_t expect_symbols)
    {
        s.start_bit = start_bit;
        s.n = 0;
        s.ends.clear();
        s.ok = s.eof = s.hit_cap = false;
        if (!s.buf) {
            s.cap = WINDOW + std::max<size_t>(expect_symbols, 1u << 16);
            s.buf = alloc_symbols(s.cap);
            if (!s.buf) {
                s.error = "out of memory";
                return Status::FAILED;
            }
            for (size_t i = 0; i < WINDOW; ++i) s.buf[i] = (uint16_t)(MARK + i);
        }
        BitReader br(data, len);
        const uint64_t total_bits = (uint64_t)len * 8;
        size_t floor = 0; // lowest buffer index a match may start at (0: the marker prefix is fair game)
        if (at_header) {
            const size_t body = gzip_header(data, len, (size_t)(start_bit >> 3));
            if (!body) {
                s.error = "not a gzip member header";
                return Status::FAILED;
            }
            br.seek((uint64_t)body * 8);
            floor = WINDOW;
        } else br.seek(start_bit);
        bool first = true;
        auto fail = [&](const char* what) {
            s.error = what;
            return first && first_block_probe ? Status::BAD_FIRST_BLOCK : Status::FAILED;
        };
        for (;;) {
            br.refill();
            const uint64_t here = br.pos();
            if (here + 3 > total_bits) return fail("truncated stream");
            if (!first) {
                if ((here >= stop_bit && (br.buf & 7) == 4) || s.n >= soft_cap) {
                    s.hit_cap = !(here >= stop_bit && (br.buf & 7) == 4);
                    s.end_bit = here;
                    s.ok = true;
                    return Status::OK;
                }
            }
            const unsigned bfinal = br.take(1), btype = br.take(2);
            if (btype == 3) return fail("reserved block type");
            if (btype == 0) {
                br.align_byte();
                br.refill();
                const uint32_t v = br.take(32);
                const unsigned ln = v & 0xFFFF;
                if ((ln ^ (v >> 16)) != 0xFFFF) return fail("stored block length check");
                const uint64_t at = br.pos(); // byte aligned
                if (at + (uint64_t)ln * 8 > total_bits) return fail("truncated stored block");
                if (!grow(s, ln + 8)) return fail("out of memory");
                const uint8_t* src = data + (at >> 3);
                uint16_t* out = s.buf + WINDOW + s.n;
                for (unsigned i = 0; i < ln; ++i) out[i] = src[i];
                s.n += ln;
                br.seek(at + (uint64_t)ln * 8);
            } else {
                const Tables* t = &fixed_tables();
                if (btype == 2) {
                    uint8_t lens[320];
                    unsigned hlit, hdist;
                    if (!read_dynamic_lengths(br, lens, hlit, hdist)) return fail("invalid dynamic block header");
                    if (!build_table(lens, hlit, LIT_BITS, dyn.lit, LIT_TABLE, false, lit_entry)) return fail("invalid literal/length code");
                    if (!build_table(lens + hlit, hdist, DIST_BITS, dyn.dist, DIST_TABLE, true, dist_entry)) return fail("invalid distance code");
                    t = &dyn;
                }
                if (br.pos() > total_bits) return fail("truncated stream");
                const char* err = inflate_block(s, br, *t, floor, total_bits);
                if (err) return fail(err);
            }
            if (s.n > HARD_CAP) return fail("block too long for the parallel inflater");
            first = false;
            if (bfinal) {
                br.align_byte();
                br.refill();
                const uint64_t at = br.pos();
                if (at + 64 > total_bits) return fail("truncated gzip trailer");
                const uint32_t crc = br.take(32);
                br.refill();
                const uint32_t isize = br.take(32);
                s.ends.push_back(MemberEnd { s.n, crc, isize });
                size_t next = (size_t)((at + 64) >> 3);
                while (next < len && data[next] == 0) ++next; // zero padding between / after members (gzip ignores it too)
                const size_t body = next < len ? gzip_header(data, len, next) : 0;
                if (!body) { // the end (trailing bytes that are no member are ignored, as gzip does with a warning)
                    s.end_bit = total_bits;
                    s.eof = true;
                    s.ok = true;
                    return Status::OK;
                }
                br.seek((uint64_t)body * 8);
                floor = WINDOW + s.n;
            }
        }
    }

    // one block's symbols; nullptr on success
    const char* inflate_block(Segment& s, BitReader& br, const Tables& t, size_t floor, uint64_t total_bits)
    {
        constexpr uint64_t LIT_MASK = (1u << LIT_BITS) - 1, DIST_MASK = (1u << DIST_BITS) - 1;
        for (;;) {
            if (!grow(s, 4096 + 258 + 8)) return "out of memory";
            uint16_t* const begin = s.buf;
            uint16_t* out = begin + WINDOW + s.n;
            uint16_t* const out_end = begin + s.cap - (258 + 8);
            const uint16_t* const lowest = begin + floor;
            bool done = false;
            const char* err = nullptr;
            while (out < out_end) {
                br.refill();
                Entry e = t.lit[br.buf & LIT_MASK];
                if (kind(e) == K_LINK) e = t.lit[e.val + ((br.buf >> LIT_BITS) & ((1u << extra(e)) - 1))];
                br.consume(e.bits);
                if (kind(e) == K_LIT) {
                    *out++ = e.val;
                    // a second literal out of the same refill (56 - 15 bits are left)
                    Entry e2 = t.lit[br.buf & LIT_MASK];
                    if (kind(e2) == K_LIT) {
                        br.consume(e2.bits);
                        *out++ = e2.val;
                    }
                    continue;
                }
                if (kind(e) == K_EOB) {
                    done = true;
                    break;
                }
                if (kind(e) != K_BASE) {
                    err = "invalid literal/length code";
                    break;
                }
                const unsigned length = e.val + (unsigned)(br.buf & ((1u << extra(e)) - 1));
                br.consume(extra(e));
                Entry d = t.dist[br.buf & DIST_MASK];
                if (kind(d) == K_LINK) d = t.dist[d.val + ((br.buf >> DIST_BITS) & ((1u << extra(d)) - 1))];
                br.consume(d.bits);
                if (kind(d) != K_BASE) {
                    err = "invalid distance code";
                    break;
                }
                const size_t dist = d.val + (size_t)(br.buf & ((1u << extra(d)) - 1));
                br.consume(extra(d));
                if (dist > (size_t)(out - lowest)) {
                    err = "distance reaches before the start of the member";
                    break;
                }
                const uint16_t* src = out - dist;
                if (dist >= 4) { // four symbols per step (may write up to three symbols past the match: room is reserved)
                    uint16_t* o = out;
                    uint16_t* const oe = out + length;
                    do {
                        std::memcpy(o, src, 8);
                        o += 4;
                        src += 4;
                    } while (o < oe);
                } else {
                    for (unsigned i = 0; i < length; ++i) out[i] = src[i];
                }
                out += length;
                if (br.pos() > total_bits) {
                    err = "truncated stream";
                    break;
                }
            }
            s.n = (size_t)(out - begin) - WINDOW;
            if (err) return err;
            if (done) return br.pos() > total_bits ? "truncated stream" : nullptr;
            if (br.pos() > total_bits) return "truncated stream";
            if (s.n > HARD_CAP) return "block too long for the parallel inflater";
        }
    }

    // first position >= from_bit (and < to_bit) that looks like the header of a dynamic, non-final block and whose block
    // inflates; the segment is then inflated from there like run().  Returns false if there is none.
    bool find_and_run(Segment& s, uint64_t from_bit, uint64_t to_bit, uint64_t stop_bit, size_t soft_cap, size_t expect_symbols)
    {
        const uint64_t total_bits = (uint64_t)len * 8;
        if (to_bit + 80 > total_bits) to_bit = total_bits > 80 ? total_bits - 80 : 0;
        for (uint64_t b = from_bit; b < to_bit; ++b) {
            const uint8_t* p = data + (b >> 3);
            uint64_t w;
            if ((size_t)(data + len - p) >= 8) std::memcpy(&w, p, 8);
            else break;
            w >>= (b & 7);
            if ((w & 7) != 4) continue;                                // BFINAL 0, BTYPE 2
            if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue; // HLIT, HDIST
            const unsigned hclen = (unsigned)((w >> 13) & 15) + 4;
            {
                // complete code-length code?  (19 x 3 bits from bit 17 on)
                BitReader br(data, len);
                br.seek(b + 17);
                int left = 128; // in units of 2^-7
                bool over = false;
                for (unsigned i = 0; i < hclen; ++i) {
                    if (br.n < 3) br.refill();
                    const unsigned l = br.take(3);
                    if (l) left -= 128 >> l;
                    if (left < 0) {
                        over = true;
                        break;
                    }
                }
                if (over || left != 0) continue;
            }
            {
                BitReader br(data, len);
                br.seek(b + 3);
                uint8_t lens[320];
                unsigned hlit, hdist;
                if (!read_dynamic_lengths(br, lens, hlit, hdist)) continue;
                // both codes complete (a lone distance code or none at all is legal, and zlib does emit those)
                auto complete = [](const uint8_t* l, unsigned n, bool lenient) {
                    int left = 1 << 15, codes = 0, maxl = 0;
                    for (unsigned i = 0; i < n; ++i)
                        if (l[i]) {
                            left -= 1 << (15 - l[i]);
                            ++codes;
                            maxl = std::max<int>(maxl, l[i]);
                        }
                    if (left == 0) return true;
                    return lenient && left > 0 && (codes == 0 || (codes == 1 && maxl == 1));
                };
                if (!complete(lens, hlit, false) || !complete(lens + hlit, hdist, true)) continue;
            }
            const Status st = run(s, b, false, stop_bit, soft_cap, true, expect_symbols);
            if (st == Status::OK) return true;
            if (st == Status::FAILED) return false; // (a later block failed: the stitcher inflates this range again and reports)
        }
        return false;
    }
};

uint32_t (*fast_crc32())(uint32_t, const void*, size_t)
{
    static uint32_t (*fn)(uint32_t, const void*, size_t) = [] {
        typedef uint32_t (*F)(uint32_t, const void*, size_t);
        if (std::getenv("DRPRG_HIP_NO_LIBDEFLATE")) return (F) nullptr;
        void* h = nullptr;
        for (const char* name : { "libdeflate.so.0", "libdeflate.so" })
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        return h ? reinterpret_cast<F>(dlsym(h, "libdeflate_crc32")) : (F) nullptr;
    }();
    return fn;
}

inline uint32_t crc_of(const void* p, size_t n)
{
    if (auto f = fast_crc32()) return f(0, p, n);
    return (uint32_t)crc32_z(0L, (const Bytef*)p, n);
}

} // namespace

struct ParallelGunzip::Impl {
    const uint8_t* data;
    size_t len;
    int threads;
    size_t chunk;
    uint64_t next_bit = 0;        // where the stitched stream stands (a block boundary, or a member header when at_header)
    bool at_header = true, done = false;
    size_t next_chunk = 0;        // the chunk whose range [next_chunk * chunk, ...) comes next
    // the rounds run on a producer thread, up to a round ahead of read(): inflating the next chunks overlaps resolving
    // and parsing the previous ones
    std::deque<std::unique_ptr<Segment>> ready; // guarded by mu, like done_pub / failed / stop
    std::mutex mu;
    std::condition_variable cv;
    std::thread producer;
    std::unique_ptr<Crew> round_crew, read_crew; // (one each: a round runs while read() resolves the round before it)
    bool started = false, done_pub = false, stop = false;
    std::string failed;
    std::vector<uint8_t> window = std::vector<uint8_t>(WINDOW, 0); // the 32 KB before the next segment to be stitched
    uint32_t run_crc = 0;
    uint64_t run_len = 0;
    std::atomic<uint64_t> accepted { 0 }, redone { 0 };
    double t_read = 0;
    double t_decode = 0, t_stitch = 0, t_resolve = 0, t_wait = 0; // wall seconds (DRPRG_GZ_DEBUG=1 prints them)
    uint64_t n_rounds = 0, n_reads = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // symbol buffers go round: a fresh 20 MB allocation per chunk is 5000 page faults, serialised between the threads by the kernel
    std::mutex pool_mu;
    std::vector<std::pair<uint16_t*, size_t>> pool;
    Segment* new_segment()
    {
        {
            std::lock_guard<std::mutex> g(pool_mu);
            if (!pool.empty()) {
                const auto b = pool.back();
                pool.pop_back();
                return new Segment(b.first, b.second);
            }
        }
        std::pair<uint16_t*, size_t> b;
        if (SymbolCache::get().take(b)) return new Segment(b.first, b.second);
        return new Segment;
    }
    void recycle(Segment& s)
    {
        if (!s.buf) return;
        std::lock_guard<std::mutex> g(pool_mu);
        if (pool.size() < (size_t)threads * 2 && s.cap <= 16 * (WINDOW + expect_symbols())) {
            pool.push_back({ s.buf, s.cap });
            s.buf = nullptr;
        }
    }
    ~Impl()
    {
        const double d0 = now();
        shutdown();
        const double d1 = now();
        round_crew.reset();
        read_crew.reset();
        if (std::getenv("DRPRG_GZ_DEBUG"))
            std::fprintf(stderr, "[pgunzip] inside read() %.3f s in all; closing: producer %.3f s, crews %.3f s\n", t_read, d1 - d0, now() - d1);
        if (std::getenv("DRPRG_GZ_DEBUG"))
            std::fprintf(stderr, "[pgunzip] threads %d chunk %zu: %llu rounds decode %.3f s stitch %.3f s | %llu reads resolve %.3f s wait-for-producer %.3f s | accepted %llu redone %llu\n",
                threads, chunk, (unsigned long long)n_rounds, t_decode, t_stitch, (unsigned long long)n_reads, t_resolve, t_wait,
                (unsigned long long)accepted.load(), (unsigned long long)redone.load());
        for (auto& b : pool) SymbolCache::get().give(b.first, b.second);
        for (auto& sgm : ready) // (a reader that stopped early)
            if (sgm && sgm->buf) {
                SymbolCache::get().give(sgm->buf, sgm->cap);
                sgm->buf = nullptr;
            }
    }

    size_t expect_symbols() const { return chunk * 10; } // (address space: only what is written gets pages)
    // symbols after which a chunk's inflater stops even without having reached the next chunk (the stitcher goes on from there);
    // DRPRG_GZ_SOFT_CAP: a small one, for the tests of that path
    size_t soft_cap() const
    {
        if (const char* e = std::getenv("DRPRG_GZ_SOFT_CAP"))
            if (const long v = std::atol(e); v > 0) return (size_t)v;
        return std::max<size_t>(chunk * 48, size_t(1) << 24);
    }

    // one round: `threads` chunks inflated at once, then stitched onto `ready`
    void round()
    {
        const size_t n_chunks = (len + chunk - 1) / chunk;
        const size_t first = next_chunk, last = std::min(n_chunks, first + (size_t)threads);
        std::vector<std::unique_ptr<Segment>> seg(last - first);
        std::atomic<size_t> cursor { 0 };
        const uint64_t stand = next_bit;
        const bool stand_header = at_header;
        auto job = [&]() {
            try {
                std::unique_ptr<Inflater> inf(new Inflater { data, len, {} });
                for (size_t j; (j = cursor.fetch_add(1)) < seg.size();) {
                    const size_t c = first + j;
                    seg[j].reset(new_segment());
                    const uint64_t lo = (uint64_t)c * chunk * 8, hi = (uint64_t)(c + 1) * chunk * 8;
                    if (j == 0) { // the stream stands at a known position: no search
                        if (inf->run(*seg[j], stand, stand_header, hi, soft_cap(), false, expect_symbols()) != Status::OK) seg[j]->ok = false;
                    } else if (stand > lo || !inf->find_and_run(*seg[j], lo, hi, hi, soft_cap(), expect_symbols())) seg[j]->ok = false;
                }
            } catch (const std::exception&) { // (out of memory: the chunks this thread did not finish are inflated again by the stitcher)
            }
        };
        const int nt = (int)std::min<size_t>((size_t)threads, seg.size());
        const double t0 = now();
        if (!round_crew) round_crew.reset(new Crew(threads - 1));
        round_crew->run(job, nt);
        const double t1 = now();
        t_decode += t1 - t0;
        ++n_rounds;
        struct Tick {
            double& acc;
            double from;
            ~Tick() { acc += now() - from; }
        } tick { t_stitch, t1 };
        // ---- stitch ----
        std::unique_ptr<Inflater> inf;
        for (size_t j = 0; j < seg.size() && !done; ++j) {
            const size_t c = first + j;
            const uint64_t hi = (uint64_t)(c + 1) * chunk * 8;
            if (next_bit >= hi && !at_header) continue; // an earlier segment ran past this chunk (long blocks): nothing to do here
            bool again = true;
            while (again && !done) {
                again = false;
                std::unique_ptr<Segment> s;
                if (seg[j] && seg[j]->ok && seg[j]->start_bit == next_bit && (j == 0 || !at_header)) {
                    s = std::move(seg[j]);
                    ++accepted;
                } else {
                    if (!inf) inf.reset(new Inflater { data, len, {} });
                    s.reset(new_segment());
                    if (inf->run(*s, next_bit, at_header, hi, soft_cap(), false, expect_symbols()) != Status::OK)
                        throw Error(DRPRG_EIO, "corrupt gzip stream (" + s->error + ")");
                    ++redone;
                    if (seg[j]) recycle(*seg[j]);
                    seg[j].reset();
                }
                // (a member that ends inside a segment restarts at a header the segment itself skipped: the stream stands mid-member
                // again at its end, except at the very end of the file)
                next_bit = s->end_bit;
                at_header = false;
                if (s->eof) done = true;
                if (s->hit_cap) again = true; // the chunk goes on from where the cap stopped this segment
                s->window = window;
                advance_window(*s);
                publish(std::move(s));
            }
        }
        for (auto& u : seg)
            if (u) recycle(*u);
        next_chunk = last;
        if (next_chunk >= n_chunks && !done) {
            // chunks are exhausted but the stream has not ended: the tail (after the last chunk's stop position) in one go
            Inflater tail { data, len, {} };
            while (!done) {
                std::unique_ptr<Segment> s(new_segment());
                if (tail.run(*s, next_bit, at_header, ~uint64_t(0), soft_cap(), false, expect_symbols()) != Status::OK)
                    throw Error(DRPRG_EIO, "corrupt gzip stream (" + s->error + ")");
                ++redone;
                next_bit = s->end_bit;
                at_header = false;
                if (s->eof) done = true;
                s->window = window;
                advance_window(*s);
                publish(std::move(s));
            }
        }
    }

    void publish(std::unique_ptr<Segment> s)
    {
        std::lock_guard<std::mutex> g(mu);
        ready.push_back(std::move(s));
        cv.notify_all();
    }

    void produce()
    {
        try {
            while (!done) {
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return stop || ready.size() < (size_t)threads; });
                    if (stop) return;
                }
                round();
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> g(mu);
            failed = e.what();
        }
        std::lock_guard<std::mutex> g(mu);
        done_pub = true;
        cv.notify_all();
    }

    void shutdown()
    {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
            cv.notify_all();
        }
        if (producer.joinable()) producer.join();
    }

    // `window` := the last 32 KB of (window + the segment's symbols, resolved)
    void advance_window(const Segment& s)
    {
        const uint16_t* sym = s.buf + WINDOW;
        if (s.n >= WINDOW) {
            std::vector<uint8_t> w(WINDOW);
            for (size_t i = 0; i < WINDOW; ++i) {
                const uint16_t v = sym[s.n - WINDOW + i];
                w[i] = v < MARK ? (uint8_t)v : s.window[v - MARK];
            }
            window.swap(w);
        } else {
            std::vector<uint8_t> w(WINDOW);
            std::memcpy(w.data(), s.window.data() + s.n, WINDOW - s.n);
            for (size_t i = 0; i < s.n; ++i) {
                const uint16_t v = sym[i];
                w[WINDOW - s.n + i] = v < MARK ? (uint8_t)v : s.window[v - MARK];
            }
            window.swap(w);
        }
    }

    struct Job {
        const Segment* seg;
        size_t from, count;
        char* dst;
        uint32_t crc = 0;
        int member_end = -1; // index into seg->ends if a member ends right after this job
    };

    size_t read(char* dst, size_t cap)
    {
        struct Tock {
            double& acc;
            double from;
            ~Tock() { acc += now() - from; }
        } tock { t_read, now() };
        std::vector<Job> jobs;
        std::vector<std::unique_ptr<Segment>> used; // segments handed out completely: freed when the jobs are done
        size_t total = 0;
        constexpr size_t PIECE = size_t(1) << 18; // (a read ends on a barrier: several pieces per thread even out who is late)
        if (!started) {
            started = true;
            producer = std::thread([this] { produce(); });
        }
        for (;;) {
            std::unique_lock<std::mutex> g(mu);
            if (ready.empty()) {
                if (total > 0) break; // (hand over what there is)
                const double w0 = now();
                cv.wait(g, [&] { return !ready.empty() || done_pub; });
                t_wait += now() - w0;
                if (ready.empty()) {
                    if (!failed.empty()) throw Error(DRPRG_EIO, failed);
                    break;
                }
            }
            if (ready.front()->read_pos == ready.front()->n && ready.front()->ends_done == ready.front()->ends.size()) {
                // a segment without text and without a member end (a chunk that was inflated again and held nothing but an
                // empty stored block, what a flush leaves): nothing to hand over -- and it must not look like "dst is full"
                used.push_back(std::move(ready.front()));
                ready.pop_front();
                cv.notify_all();
                continue;
            }
            Segment& s = *ready.front();
            // the next unchecked member end bounds the piece: a CRC never runs across it
            const bool has_end = s.ends_done < s.ends.size();
            const size_t bound = has_end ? s.ends[s.ends_done].at : s.n;
            const size_t take = std::min(bound - s.read_pos, cap - total);
            const bool reaches_end = has_end && s.read_pos + take == bound;
            if (take == 0 && !reaches_end) break; // dst is full
            size_t off = 0;
            do { // pieces of at most 1 MB: the threads share the work evenly; a zero-length job checks an empty member
                const size_t c = std::min(PIECE, take - off);
                Job j { &s, s.read_pos + off, c, dst + total + off };
                if (reaches_end && off + c == take) j.member_end = (int)s.ends_done;
                jobs.push_back(j);
                off += c;
            } while (off < take);
            s.read_pos += take;
            total += take;
            if (reaches_end) ++s.ends_done;
            if (s.read_pos == s.n && s.ends_done == s.ends.size()) {
                used.push_back(std::move(ready.front()));
                ready.pop_front();
                cv.notify_all();
            }
        }
        if (jobs.empty()) return 0;
        std::atomic<size_t> cursor { 0 };
        auto work = [&]() {
            for (size_t i; (i = cursor.fetch_add(1)) < jobs.size();) {
                Job& j = jobs[i];
                const uint16_t* sym = j.seg->buf + WINDOW + j.from;
                const uint8_t* w = j.seg->window.data();
                char* o = j.dst;
                size_t k = 0;
                for (; k + 16 <= j.count; k += 16) { // sixteen symbols at a time when none of them is a marker
                    const __m128i a = _mm_loadu_si128((const __m128i*)(sym + k)), b = _mm_loadu_si128((const __m128i*)(sym + k + 8));
                    if (_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) { // a sign bit of a 16-bit lane: a marker
                        for (size_t q = k; q < k + 16; ++q) {
                            const uint16_t v = sym[q];
                            o[q] = (char)(v < MARK ? (uint8_t)v : w[v - MARK]);
                        }
                    } else _mm_storeu_si128((__m128i*)(o + k), _mm_packus_epi16(a, b));
                }
                for (; k < j.count; ++k) {
                    const uint16_t v = sym[k];
                    o[k] = (char)(v < MARK ? (uint8_t)v : w[v - MARK]);
                }
                j.crc = j.count ? crc_of(o, j.count) : 0;
            }
        };
        const int nt = (int)std::min<size_t>((size_t)threads, (jobs.size() + 1) / 2);
        const double r0 = now();
        if (!read_crew) read_crew.reset(new Crew(threads - 1));
        read_crew->run(work, nt);
        t_resolve += now() - r0;
        ++n_reads;
        for (auto& u : used) recycle(*u);
        for (const Job& j : jobs) {
            if (j.count) {
                run_crc = (uint32_t)crc32_combine(run_crc, j.crc, (z_off_t)j.count);
                run_len += j.count;
            }
            if (j.member_end >= 0) {
                const MemberEnd& e = j.seg->ends[(size_t)j.member_end];
                if (run_crc != e.crc || (uint32_t)run_len != e.isize) throw Error(DRPRG_EIO, "gzip member fails its CRC-32 / length check");
                run_crc = 0;
                run_len = 0;
            }
        }
        return total;
    }
};

ParallelGunzip::ParallelGunzip(const unsigned char* data, size_t len, int threads, size_t chunk_bytes) : impl_(new Impl)
{
    impl_->data = data;
    impl_->len = len;
    impl_->threads = std::max(1, threads);
    // (every chunk in flight holds ~12x its size in symbols, memory the kernel has to zero first: 1 MB chunks are as fast as 4 MB ones)
    if (!chunk_bytes) chunk_bytes = std::min<size_t>(size_t(1) << 20, std::max<size_t>(size_t(128) << 10, len / ((size_t)impl_->threads * 8)));
    impl_->chunk = std::max<size_t>(chunk_bytes, 1024);
    if (!gzip_header(data, len, 0)) throw Error(DRPRG_EIO, "not a gzip file");
}

ParallelGunzip::~ParallelGunzip() = default;
size_t ParallelGunzip::read(char* dst, size_t cap) { return impl_->read(dst, cap); }
uint64_t ParallelGunzip::chunks_accepted() const { return impl_->accepted; }
uint64_t ParallelGunzip::chunks_redone() const { return impl_->redone; }

} // namespace drprg
