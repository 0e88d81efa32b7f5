// pack.h -- host side of the 2-bit packed read format (kernels.h SketchArgs::packed; SURVEY.md section 8f NEXT-4): what crosses PCIe
// when drprg_hip_set_input_format(ctx, 1) is in force.  Letter = bits 2:1 of the byte (A 0, C 1, T 2, G 3 for either case; any other
// byte gets the letter of those bits too and its position goes to `npos`), base i of the stream in bits [2 (i & 31) + 1 : 2 (i & 31)]
// of 64-bit word i >> 5 -- read as 32-bit words that is 16 bases per word, first base lowest.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace drprg {

// Appends seq[0, len) to the packed stream that holds n_bases bases so far.  words64: room for (n_bases + len) / 32 + 2 words; the
// word that holds base n_bases (if any base of it is set) must be valid below that base and ZERO above, which is how this function
// leaves it (a fresh stream: words64[0] = 0).  Positions of bytes that are not ACGTacgt are appended to npos (as n_bases + i).
void pack_append(uint64_t* words64, uint64_t& n_bases, const char* seq, size_t len, std::vector<uint64_t>& npos);

#if defined(__x86_64__)
// 32 bytes at p -> their letters (64 bits); bit i of bad: byte i is not ACGTacgt.  Shift-and-mask to the 2-bit letters, two
// multiply-adds gather four letters per byte, one byte shuffle gathers the eight bytes; the validity test in the same registers.
__attribute__((target("avx2"))) static inline uint64_t pack32_avx2(const char* p, uint32_t& bad)
{
    const __m256i b = _mm256_loadu_si256((const __m256i*)p);
    const __m256i c = _mm256_and_si256(_mm256_srli_epi16(b, 1), _mm256_set1_epi8(3)); // the letters, one per byte
    // a byte is a base iff its upper-case form is the letter its two bits stand for
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i upper = _mm256_and_si256(b, _mm256_set1_epi8((char)0xDF));
    bad = ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(upper, _mm256_shuffle_epi8(lut, c)));
    const __m256i p2 = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0401));   // byte pairs: first + 4 * second
    const __m256i p4 = _mm256_madd_epi16(p2, _mm256_set1_epi32(0x00100001)); // 16-bit pairs: first + 16 * second -> one byte of four letters per dword
    const __m256i g = _mm256_shuffle_epi8(p4,
        _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    return (uint64_t)(uint32_t)_mm256_extract_epi32(g, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(g, 4) << 32);
}

// pack_append for a caller that knows (a) the CPU has AVX2 and (b) the 31 bytes behind seq[len - 1] are readable (the parser's fast
// path: the '+' and quality lines follow): the last, partial 32 bases take one masked vector step instead of a byte loop -- with
// 150-base reads that loop was 22 of every 150 bases and made packing cost more than the memcpy it replaces.
__attribute__((target("avx2"))) static inline void pack_append_overread_avx2(uint64_t* words64, uint64_t& n_bases, const char* seq, size_t len,
    std::vector<uint64_t>& npos)
{
    const unsigned shift = (unsigned)(n_bases & 31) * 2;
    uint64_t idx = n_bases >> 5;
    uint64_t acc = shift ? words64[idx] : 0;
    for (size_t i = 0; i < len; i += 32) {
        const size_t r = len - i < 32 ? len - i : 32;
        uint32_t bad;
        uint64_t v = pack32_avx2(seq + i, bad);
        if (r < 32) {
            v &= (1ull << (2 * r)) - 1;
            bad &= (1u << r) - 1;
        }
        if (__builtin_expect(bad != 0, 0))
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

} // namespace drprg
