// common.h -- shared host-side types and the k-mer hash for the MI355X predict hot path.
//
// The hash and base encoding follow pandora's KmerHash / inthash (external program invoked at
// /root/reference/src/lib.rs:580-642); see DESIGN.md "Semantics".
#pragma once
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace drprg {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// errno-style negative return codes of the C ABI (include/drprg_hip.h)
enum : int {
    DRPRG_OK = 0,
    DRPRG_EINVAL = -22,
    DRPRG_ENOENT = -2,
    DRPRG_EIO = -5,
    DRPRG_ENOMEM = -12,
    DRPRG_ENODEV = -19,
    DRPRG_EOVERFLOW = -75,
    DRPRG_EFORMAT = -84,
    DRPRG_ENODATA = -61, // the reads of the sample are not (all) resident in HBM: the caller reads the file instead
    DRPRG_EAGAIN_SERIAL = -11, // internal: the parallel ingest hands the file to the serial reader (nothing was mapped yet)
};

inline uint64_t kmer_mask(int k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }

inline uint64_t hash64(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

inline int nt4(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

// Multipliers of the two Bloom-filter levels shared by FlatIndex::build (index.cpp) and sketch_filter_kernel
// (sketch_filter.hip, where the layout is described); level 1 is a 24 x 24 bit multiply.
constexpr uint32_t BLOOM_C0 = 0xC2B2AFu; // level 0, also 24 x 24 bit
constexpr uint32_t BLOOM_C1 = 0x9E3779u;
constexpr uint32_t BLOOM_C2 = 0x85EBCA6Bu;
constexpr uint32_t BLOOM_CR = 0xC2B2AE35u; // second stage of the level-0 form
constexpr uint32_t BLOOMR_WBITS = 14;      // 64 KB
// Middle tier of the filter (k = 15, indexes too large for the all-LDS form; sketch_filter.hip): the 12-mers of a group of four
// positions are keyed CANONICALLY -- min(code, reverse complement of the code), which halves the entries of both tables --,
// tested against the 128 KB level-0 array in LDS and then against an exact 2^24-bit bitmap (2 MB, L2-resident); the four 15-mer
// codes of a surviving group against ONE 16-byte block of a split-block Bloom filter in global memory (L2-resident as well): the
// block is selected by the group's canonical 12-mer (the same hash as level 0), every index code is entered once per alignment
// (in the block of the 12-mer at its offset 0..3) and sets one bit in each of the block's four words (bits 31:27, 26:22, 21:17,
// 16:12 of code * BLOOM_CR).  2^midc_wbits blocks.
constexpr uint32_t MID_BITMAP_WORDS = 1u << 19;
// largest index (records) the middle tier serves: beyond it hashing every k-mer (sketch_wave_kernel) is faster -- measured crossover
// between 121 k records (2.5 against 3.7 ms per 10 M reads) and 244 k (5.4 against 4.2 ms); DRPRG_MID_MAX_RECORDS moves it
inline size_t mid_tier_max_records()
{
    if (const char* e = std::getenv("DRPRG_MID_MAX_RECORDS")) return (size_t)std::strtoull(e, nullptr, 10);
    return 180000;
}
constexpr uint32_t MID_C_MAX_WBITS = 17; // 2^17 blocks of 16 bytes = 2 MB
// reverse complement of a 12-mer code (2 bits per base, first base in the lowest bits, alphabet A 0, C 1, T 2, G 3: complement = ^ 2)
inline uint32_t rc12_code(uint32_t x)
{
    uint32_t r = 0;
    for (int i = 0; i < 12; ++i) r |= (((x >> (2 * i)) & 3u) ^ 2u) << (2 * (11 - i));
    return r;
}
inline uint32_t canon12_code(uint32_t x)
{
    x &= 0xFFFFFFu;
    const uint32_t r = rc12_code(x);
    return x < r ? x : r;
}

// canonical hash of a k-mer given as a string of exactly k ACGT characters.
// strand = true when the forward k-mer hashes <= its reverse complement.
inline bool canonical_kmer_hash(const char* s, int k, uint64_t& h, bool& strand)
{
    const uint64_t mask = kmer_mask(k);
    uint64_t f = 0, r = 0;
    for (int i = 0; i < k; ++i) {
        int c = nt4((unsigned char)s[i]);
        if (c > 3) return false;
        f = (f << 2) | (uint64_t)c;
        r = (r >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
    }
    uint64_t hf = hash64(f & mask, mask), hr = hash64(r & mask, mask);
    h = hf < hr ? hf : hr;
    strand = hf <= hr;
    return true;
}

// Mapping parameters of one `pandora map` / `discover` invocation
// (argv built at /root/reference/src/predict.rs:236-245 and src/lib.rs:594-618).
struct MapParams {
    int w = 11;
    int k = 15;
    int max_diff = 250;             // --max-diff; 2k+1 with -I
    double error_rate = 0.11;       // -e; 0.001 with -I
    uint32_t min_cluster_size = 10; // -c
    bool illumina = false;          // -I
    uint64_t genome_size = 5000000; // -g
    double genotyping_error_rate = 0.01;
    bool binomial = false;          // --bin: binomial model of the k-mer coverages (default: negative binomial)
    int kernel_mode = 0; // 0 auto, 1 direct sketch (sketch_probe_kernel) + generic cluster pipeline, 2 Bloom-prefiltered
                         // (sketch_filter_kernel), 3 direct sketch in its candidate form + read_cluster_kernel
    double cluster_fraction() const; // 0.5 / exp(error_rate * k)
};

} // namespace drprg
