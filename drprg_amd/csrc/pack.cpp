// pack.cpp -- see pack.h.  32 bases per step with AVX2 (shift-and-mask to the 2-bit letters, two multiply-adds gather four letters
// per byte, one byte shuffle gathers the eight bytes), the validity test in the same registers; scalar for tails and other CPUs.
#include "pack.h"
#include <cstdlib>

namespace drprg {

namespace {

inline bool is_acgt(unsigned char c)
{
    const unsigned char u = c & 0xDFu;
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

// up to 32 bases -> their letters in the low 2 * n bits; bit i of bad: byte i is not ACGTacgt
inline uint64_t pack_scalar(const char* p, size_t n, uint32_t& bad)
{
    uint64_t v = 0;
    bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)p[i];
        v |= (uint64_t)((c >> 1) & 3u) << (2 * i);
        if (!is_acgt(c)) bad |= 1u << i;
    }
    return v;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void pack_append_avx2(uint64_t* words64, uint64_t& n_bases, const char* seq, size_t len, std::vector<uint64_t>& npos)
{
    const unsigned shift = (unsigned)(n_bases & 31) * 2;
    uint64_t idx = n_bases >> 5;
    uint64_t acc = shift ? words64[idx] : 0;
    size_t i = 0;
    for (; i + 32 <= len; i += 32) {
        uint32_t bad;
        const uint64_t v = pack32_avx2(seq + i, bad);
        if (bad)
            for (uint32_t m = bad; m; m &= m - 1) npos.push_back(n_bases + i + (uint64_t)__builtin_ctz(m));
        acc |= v << shift;
        words64[idx++] = acc;
        acc = shift ? v >> (64 - shift) : 0;
    }
    if (i < len) {
        const size_t r = len - i;
        uint32_t bad;
        const uint64_t v = pack_scalar(seq + i, r, bad);
        if (bad)
            for (uint32_t m = bad; m; m &= m - 1) npos.push_back(n_bases + i + (uint64_t)__builtin_ctz(m));
        acc |= v << shift;
        if (shift + 2 * r >= 64) {
            words64[idx++] = acc;
            acc = shift ? v >> (64 - shift) : 0;
        }
    }
    words64[idx] = acc;
    n_bases += len;
}
#endif

void pack_append_scalar(uint64_t* words64, uint64_t& n_bases, const char* seq, size_t len, std::vector<uint64_t>& npos)
{
    uint64_t idx = n_bases >> 5;
    unsigned shift = (unsigned)(n_bases & 31) * 2;
    uint64_t acc = shift ? words64[idx] : 0;
    for (size_t i = 0; i < len; ++i) {
        const unsigned char c = (unsigned char)seq[i];
        acc |= (uint64_t)((c >> 1) & 3u) << shift;
        if (!is_acgt(c)) npos.push_back(n_bases + i);
        shift += 2;
        if (shift == 64) {
            words64[idx++] = acc;
            acc = 0;
            shift = 0;
        }
    }
    words64[idx] = acc;
    n_bases += len;
}

} // namespace

void pack_append(uint64_t* words64, uint64_t& n_bases, const char* seq, size_t len, std::vector<uint64_t>& npos)
{
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && !std::getenv("DRPRG_PARSE_NO_SIMD");
    if (have_avx2) {
        pack_append_avx2(words64, n_bases, seq, len, npos);
        return;
    }
#endif
    pack_append_scalar(words64, n_bases, seq, len, npos);
}

} // namespace drprg
