// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the drprg predict hot path.
//
//   K1+K2  sketch_probe_kernel   (w,k)-minimizer sketch of a batch of reads + open-addressed
//                                minimizer -> PRG-k-mer-node probe, emitting hits
//   K3a-e  cluster kernels       per-read hit clustering, size / overlap filters, and atomic
//                                accumulation into the per-k-mer-node fwd/rev coverage vector
//
// They replace, inside the external `pandora map` / `pandora discover` process that
// /root/reference/src/lib.rs:513-642 spawns: Seq::minimizer_sketch, add_read_hits,
// define_clusters, filter_clusters, add_clusters_to_pangraph and add_hits_to_kmergraphs
// (SURVEY.md section 8, rows a-5..a-8).  Integer / indexing work: no MFMA.
//
// Layout in HBM
//   bases   u8[n_bases]      all reads of the batch back to back (ASCII), 16-byte aligned
//   offsets u64[n_reads+1]   read i = bases[offsets[i], offsets[i+1])
//   table   open addressed, 2^bits slots: slot_key (u32 when k<=15, u64 otherwise), slot_rec {off,cnt}
//   records u32 rec_knode (global k-mer node id << 1 | strand), u16 rec_prg
//   covg    u32[2*n_knodes]  [2g] forward, [2g+1] reverse
//
// The sketch kernel does not look at reads one by one: it tiles the *concatenated* base buffer, so
// it is load balanced for any read-length mix; read starts are injected as flags on the staged bases,
// which makes a k-mer that would straddle two reads invalid exactly like one holding an N.
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace drprg {
namespace dev {

// ---------------------------------------------------------------------------------------------
// hash
// ---------------------------------------------------------------------------------------------
template <typename HT> struct HashTraits;
template <> struct HashTraits<uint32_t> {
    static constexpr uint32_t EMPTY = 0xFFFFFFFFu; // never a hash: this path serves k <= 15 (30 bits)
    // minimap hash64 restricted to <= 30 bits: every intermediate is taken mod 2^(2k), so 32-bit
    // arithmetic is exact; the final (key + key<<31) term vanishes below 31 bits.
    __device__ static inline uint32_t mix(uint32_t key, uint32_t mask)
    {
        key = (~key + (key << 21)) & mask;
        key = key ^ (key >> 24);
        key = (key + (key << 3) + (key << 8)) & mask;
        key = key ^ (key >> 14);
        key = (key + (key << 2) + (key << 4)) & mask;
        key = key ^ (key >> 28);
        return key;
    }
};
template <> struct HashTraits<uint64_t> {
    static constexpr uint64_t EMPTY = ~0ULL; // k <= 31: hashes stay below 2^62
    __device__ static inline uint64_t mix(uint64_t key, uint64_t mask)
    {
        key = (~key + (key << 21)) & mask;
        key = key ^ (key >> 24);
        key = (key + (key << 3) + (key << 8)) & mask;
        key = key ^ (key >> 14);
        key = (key + (key << 2) + (key << 4)) & mask;
        key = key ^ (key >> 28);
        key = (key + (key << 31)) & mask;
        return key;
    }
};

__device__ inline uint32_t table_slot_dev(uint64_t key, uint32_t bits)
{
    return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - bits));
}

// ASCII -> code byte: bits 0-1 base (A0 C1 G2 T3), bit 2 = not ACGT
__device__ inline uint32_t encode_base(uint32_t c)
{
    uint32_t u = c & 0xDFu; // upper case
    uint32_t x = (u >> 1) & 3u;
    x ^= x >> 1;
    bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
    return ok ? x : 4u;
}
// four bases at once (SWAR): same mapping on every byte of a dword
__device__ inline uint32_t encode4(uint32_t word)
{
    const uint32_t u = word & 0xDFDFDFDFu; // upper case
    const uint32_t sel = (u >> 1) & 0x03030303u; // A0 C1 T2 G3: a byte-wise index into two 4-entry tables
    // v_perm_b32 as a 4-entry byte LUT: selector bytes 0..3 pick bytes of the second operand
    const uint32_t code = __builtin_amdgcn_perm(0u, 0x02030100u, sel);   // -> A0 C1 G2 T3
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, sel); // the letter that index stands for
    const uint32_t diff = u ^ expect; // a byte is a valid base iff it equals the letter of its index
    const uint32_t bad = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u; // bit 7 set in every non-zero byte
    return code | (bad >> 5); // 0x80 >> 5 = 4: the "not ACGT" flag
}

// read holding global base position gp, searched upwards from read `lo` (offsets[lo] <= gp)
__device__ inline uint32_t find_read_from(const uint64_t* __restrict__ offsets, uint32_t n_reads, uint32_t lo, uint64_t gp)
{
    uint32_t step = 1, hi = lo + 1;
    while (hi < n_reads && offsets[hi] <= gp) { // gallop
        lo = hi;
        step <<= 1;
        hi = (n_reads - lo > step) ? lo + step : n_reads;
    }
    // invariant: offsets[lo] <= gp < offsets[hi]  (offsets[n_reads] = n_bases > gp)
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= gp) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// K1 + K2: sketch + probe
// ---------------------------------------------------------------------------------------------
constexpr int SK_THREADS = 256;
constexpr int SK_G = 16;                       // k-mer positions per thread
constexpr int SK_NPOS = SK_THREADS * SK_G;     // 4096 hashed positions per tile
constexpr int SK_CODES = SK_NPOS + 48;         // staged bases (multiple of 16 >= NPOS + k - 1, k <= 31)
constexpr int SK_MAXSTEPS = SK_G + 30;         // bases one thread rolls over (k <= 31)

// first read that starts at or after the first staged base of every tile
__global__ void tile_first_read_kernel(const uint64_t* __restrict__ offsets, uint32_t n_reads, int t_eval, int halo,
    uint32_t n_tiles, uint32_t* __restrict__ out)
{
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_tiles) return;
    int64_t lo_pos = (int64_t)b * t_eval - halo;
    if (lo_pos < 0) lo_pos = 0;
    uint32_t lo = 0, hi = n_reads;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)offsets[mid] < lo_pos) lo = mid + 1; else hi = mid;
    }
    out[b] = lo;
}

__device__ __forceinline__ int hpad(int p) { return p + (p >> 4); } // LDS index of tile position p (row of 16 + 1 pad)

// KC / WC: compile-time k / w (0 = take them from the arguments)
template <typename HT, int KC, int WC>
__global__ __launch_bounds__(SK_THREADS) void sketch_probe_kernel(SketchArgs a)
{
    using Tr = HashTraits<HT>;
    // s_hash holds hash+1 of every valid k-mer of the tile and 0 for an invalid one
    __shared__ uint4 s_code4[SK_CODES / 16];
    __shared__ HT s_hash[SK_NPOS + SK_NPOS / 16];
    __shared__ uint16_t s_strand[SK_THREADS];
    __shared__ uint16_t s_mins[SK_NPOS];
    __shared__ uint32_t s_nmin;
    uint8_t* s_code = reinterpret_cast<uint8_t*>(s_code4);

    const int tid = threadIdx.x;
    const int k = KC ? KC : a.k;
    const int w = WC ? WC : a.w;
    const int halo = a.halo;                 // multiple of 16, >= w-1
    const int t_eval = SK_NPOS - 2 * halo;   // k-mer positions evaluated by this tile
    // tile origin in global base coordinates (a multiple of 16; negative for tile 0)
    const int64_t origin = (int64_t)blockIdx.x * t_eval - halo;
    const int64_t n_bases = (int64_t)a.n_bases;

    if (tid == 0) s_nmin = 0;

    // ---- stage bases -> codes (coalesced 16-byte loads) ----
    for (int v = tid; v < SK_CODES / 16; v += SK_THREADS) {
        int64_t g = origin + (int64_t)v * 16;
        uint4 out;
        if (g >= 0 && g + 16 <= n_bases) {
            uint4 in = *reinterpret_cast<const uint4*>(a.bases + g);
            out.x = encode4(in.x); out.y = encode4(in.y); out.z = encode4(in.z); out.w = encode4(in.w);
        } else {
            uint32_t tmp[4];
            for (int q = 0; q < 4; ++q) {
                uint32_t wd = 0;
                for (int b = 0; b < 4; ++b) {
                    int64_t gg = g + q * 4 + b;
                    uint32_t c = (gg >= 0 && gg < n_bases) ? encode_base(a.bases[gg]) : 4u;
                    wd |= c << (8 * b);
                }
                tmp[q] = wd;
            }
            out.x = tmp[0]; out.y = tmp[1]; out.z = tmp[2]; out.w = tmp[3];
        }
        s_code4[v] = out;
    }
    const uint32_t first_read = a.tile_first_read[blockIdx.x];
    __syncthreads();

    // ---- phase 1: rolling canonical hash of SK_G consecutive k-mers per thread ----
    const int base0 = tid * SK_G;
    {
        const HT mask = (HT)((1ULL << (2 * k)) - 1);
        const int shift1 = 2 * (k - 1);
        const int nsteps = SK_G + k - 1;
        uint4 c0 = s_code4[tid], c1 = s_code4[tid + 1], c2 = s_code4[tid + 2];
        const uint32_t words[12] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w };
        HT fwd = 0, rev = 0;
        uint32_t strand_bits = 0, any_n = 0;
#pragma unroll
        for (int s = 0; s < SK_MAXSTEPS; ++s) {
            if (s < nsteps) {
                const uint32_t c = words[s >> 2] >> (8 * (s & 3));
                any_n |= c;
                const HT b = (HT)(c & 3u);
                fwd = ((fwd << 2) | b) & mask;
                rev = (rev >> 2) | ((b ^ (HT)3) << shift1);
                if (s >= k - 1) {
                    const int j = s - (k - 1);
                    const HT hf = Tr::mix(fwd, mask), hr = Tr::mix(rev, mask);
                    strand_bits |= (uint32_t)(hf <= hr) << j;
                    s_hash[hpad(base0 + j)] = (hf < hr ? hf : hr) + 1;
                }
            }
        }
        if (any_n & 4u) { // rare: some base of this thread's span is not ACGT (or lies outside the buffer)
            for (int j = 0; j < SK_G; ++j) {
                bool bad = false;
                for (int i = j; i < j + k; ++i) bad |= (s_code[base0 + i] & 4) != 0;
                if (bad) s_hash[hpad(base0 + j)] = 0;
            }
        }
        s_strand[tid] = (uint16_t)strand_bits;
    }
    __syncthreads();
    // ---- a k-mer must not straddle two reads: invalidate the k-1 k-mers that end inside the next read ----
    {
        const int64_t end_pos = origin + SK_CODES;
        for (uint32_t r = first_read + tid; r < a.n_reads; r += SK_THREADS) {
            const int64_t o = (int64_t)a.offsets[r];
            if (o >= end_pos) break;
            const int oc = (int)(o - origin);
            int p0 = oc - k + 1, p1 = oc < SK_NPOS ? oc : SK_NPOS;
            if (p0 < 0) p0 = 0;
            for (int p = p0; p < p1; ++p) s_hash[hpad(p)] = 0;
        }
    }
    __syncthreads();

    // ---- phase 2a: which of my positions are window minimizers? ----
    // position j is a minimizer iff some window of w consecutive valid k-mers containing j has no value below g[j].
    if (base0 >= halo && base0 < SK_NPOS - halo) {
        uint32_t minbits = 0;
        const int pbase = base0 - (w - 1); // tile position of element 0 of my neighbourhood
        if constexpr (WC > 0) {
            // branch-free: sliding minimum over windows (an invalid k-mer is 0, so an invalid window has minimum 0 and
            // never equals a valid g >= 1), then sliding maximum of the window minima over the windows holding j
            constexpr int N = SK_G + 2 * (WC - 1);
            HT m[N], own[SK_G];
#pragma unroll
            for (int i = 0; i < N; ++i) m[i] = s_hash[hpad(pbase + i)];
#pragma unroll
            for (int j = 0; j < SK_G; ++j) own[j] = m[WC - 1 + j];
            constexpr int P = (WC >= 16) ? 16 : (WC >= 8) ? 8 : (WC >= 4) ? 4 : (WC >= 2) ? 2 : 1; // largest power of two <= WC
#pragma unroll
            for (int sp = 1; sp < P; sp *= 2) {
#pragma unroll
                for (int i = 0; i + sp < N; ++i) m[i] = m[i] < m[i + sp] ? m[i] : m[i + sp];
            }
#pragma unroll
            for (int i = 0; i + WC - 1 < N; ++i) m[i] = m[i] < m[i + WC - P] ? m[i] : m[i + WC - P]; // m[i] = min g[i..i+WC-1]
            constexpr int NW = SK_G + WC - 1; // window starts that matter: 0 .. NW-1
#pragma unroll
            for (int sp = 1; sp < P; sp *= 2) {
#pragma unroll
                for (int i = 0; i + sp < NW; ++i) m[i] = m[i] > m[i + sp] ? m[i] : m[i + sp];
            }
#pragma unroll
            for (int j = 0; j < SK_G; ++j) {
                HT best = m[j] > m[j + WC - P] ? m[j] : m[j + WC - P]; // max of the minima of windows j .. j+WC-1
                minbits |= (uint32_t)(own[j] != 0 && best == own[j]) << j;
            }
        } else {
            // generic w: sequential scan that reports every window arg-min exactly once
            const int N = SK_G + 2 * (w - 1);
            HT mn = 0;
            int mpos = -1, nvalid = 0;
            for (int i = 0; i < N; ++i) {
                const HT g = s_hash[hpad(pbase + i)];
                if (g == 0) { nvalid = 0; continue; }
                ++nvalid;
                if (nvalid < w) continue;
                if (nvalid == w || mpos < i - w + 1) { // first full window after a break, or the minimum slid out
                    HT best = g;
                    for (int d = 1; d < w; ++d) {
                        HT x = s_hash[hpad(pbase + i - d)];
                        best = x < best ? x : best;
                    }
                    mn = best;
                    for (int d = w - 1; d >= 0; --d) {
                        if (s_hash[hpad(pbase + i - d)] == mn) {
                            mpos = i - d;
                            int j = i - d - (w - 1);
                            if (j >= 0 && j < SK_G) minbits |= 1u << j;
                        }
                    }
                } else if (g <= mn) {
                    mn = g;
                    mpos = i;
                    int j = i - (w - 1);
                    if (j >= 0 && j < SK_G) minbits |= 1u << j;
                }
            }
        }
        if (minbits) {
            uint32_t at = atomicAdd(&s_nmin, (uint32_t)__popc(minbits));
            while (minbits) {
                int j = __ffs(minbits) - 1;
                minbits &= minbits - 1;
                s_mins[at++] = (uint16_t)(base0 + j);
            }
        }
    }
    __syncthreads();

    // ---- phase 2b: probe the index with the compacted minimizer list ----
    const uint32_t nmin = s_nmin;
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const HT* __restrict__ slot_key = reinterpret_cast<const HT*>(a.slot_key);
    for (uint32_t i = tid; i < nmin; i += SK_THREADS) {
        const int j = s_mins[i];
        const HT h = s_hash[hpad(j)] - 1;
        uint32_t s = table_slot_dev((uint64_t)h, a.table_bits);
        bool found = false;
        while (true) {
            const HT key = slot_key[s];
            if (key == h) { found = true; break; }
            if (key == Tr::EMPTY) break;
            s = (s + 1) & tmask;
        }
        if (!found) continue;
        const uint2 rec = a.slot_rec[s];
        // a hit: locate the read and emit one hit per index record
        const uint64_t gp = (uint64_t)(origin + j);
        const uint32_t read = find_read_from(a.offsets, a.n_reads, first_read ? first_read - 1 : 0, gp);
        const uint64_t pos = gp - a.offsets[read];
        const uint32_t strand = (s_strand[j / SK_G] >> (j % SK_G)) & 1u;
        const unsigned long long at = atomicAdd(a.n_hits, (unsigned long long)rec.y);
        if (at + rec.y > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
            atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
            continue;
        }
        for (uint32_t q = 0; q < rec.y; ++q) {
            const uint32_t kn = a.rec_knode[rec.x + q]; // (global knode << 1) | strand
            const uint32_t prg = a.rec_prg[rec.x + q];
            const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u; // forward hits sort first
            a.hit_key[at + q] = pack_hit_key(read, prg, rev, (uint32_t)pos);
            a.hit_val[at + q] = kn >> 1;
        }
    }
    if (tid == 0 && nmin) atomicAdd(a.n_minimizers, (unsigned long long)nmin);
}

// ---------------------------------------------------------------------------------------------
// K1 + K2, filtered form: persistent workgroups + LDS Bloom prefilter on k-mer codes
// ---------------------------------------------------------------------------------------------
// A read minimizer can only produce a hit if its k-mer is an index k-mer (either orientation).  So instead of
// hashing every k-mer, each position's 2-bit code is tested against a Bloom filter of the index k-mer codes that
// lives in LDS for the lifetime of a persistent workgroup; only candidates get the exact treatment: canonical hash,
// exact table lookup, and -- for true index k-mers -- the window-minimizer test over the 2w-1 neighbouring hashes.
// Results are identical to sketch_probe_kernel (the Bloom filter has no false negatives); k <= 15, w <= 16.
constexpr int FT_THREADS = 512;
constexpr int FT_G = 16;
constexpr int FT_NPOS = FT_THREADS * FT_G;   // 8192 positions per tile
constexpr int FT_HALO = 16;                  // >= w-1
constexpr int FT_EVAL = FT_NPOS - 2 * FT_HALO;
constexpr int FT_CODES = FT_NPOS + 48;       // staged bases
constexpr int FT_WORDS = FT_CODES / 16;      // 515 packed words
constexpr int FT_CAND_CAP = FT_THREADS;      // one candidate per thread per round
constexpr int FT_VER_HITS = 32;              // hits verified per pass
constexpr int FT_VER_W = 31;                 // 2*16-1 neighbour slots

__global__ void tile_first_read_ft_kernel(const uint64_t* __restrict__ offsets, uint32_t n_reads, uint32_t n_tiles,
    uint32_t* __restrict__ out)
{
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_tiles) return;
    int64_t lo_pos = (int64_t)b * FT_EVAL - FT_HALO;
    if (lo_pos < 0) lo_pos = 0;
    uint32_t lo = 0, hi = n_reads;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)offsets[mid] < lo_pos) lo = mid + 1; else hi = mid;
    }
    out[b] = lo;
}

// 16 ASCII bases -> 32-bit packed 2-bit codes (first base in the top bits) + 16-bit "not ACGT" mask (bit i = base i)
__device__ inline void pack16(const uint4& in, uint32_t& packed, uint32_t& nmask)
{
    const uint32_t e0 = encode4(in.x), e1 = encode4(in.y), e2 = encode4(in.z), e3 = encode4(in.w);
    // gather the four 2-bit fields of a dword into one byte, first base highest: (x * 0x40100401) >> 24
    const uint32_t p0 = ((e0 & 0x03030303u) * 0x40100401u) >> 24, p1 = ((e1 & 0x03030303u) * 0x40100401u) >> 24;
    const uint32_t p2 = ((e2 & 0x03030303u) * 0x40100401u) >> 24, p3 = ((e3 & 0x03030303u) * 0x40100401u) >> 24;
    packed = (p0 << 24) | (p1 << 16) | (p2 << 8) | p3;
    nmask = 0;
    if ((e0 | e1 | e2 | e3) & 0x04040404u) { // rare
        // (flags * 0x01020408) >> 24 gathers the flag of byte i into bit i
        auto m4 = [](uint32_t e) { return ((((e >> 2) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; };
        nmask = m4(e0) | (m4(e1) << 4) | (m4(e2) << 8) | (m4(e3) << 12);
    }
}

// reverse complement of a k-mer code (2 bits per base, k <= 16)
__device__ inline uint32_t revcomp_code(uint32_t f, int k)
{
    uint32_t x = __brev(f);                                    // bit reversal also swaps the two bits of every base
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);   // swap them back
    return (~x) >> (32 - 2 * k);                               // complement, right-align
}

struct FtShared {
    uint32_t pack[FT_WORDS + 1];
    uint16_t nmask[FT_WORDS + 1];
    uint32_t start[(FT_CODES + 31) / 32 + 1]; // bit per staged base: a read starts here
    uint16_t cand[FT_CAND_CAP];
    uint32_t hit_slot[FT_CAND_CAP];
    uint16_t hit_pos[FT_CAND_CAP]; // bit 15 = strand of the read k-mer
    uint32_t ver[FT_VER_HITS][FT_VER_W];
    uint32_t ncand, nhit;
};

__global__ __launch_bounds__(FT_THREADS) void sketch_filter_kernel(SketchArgs a, const uint32_t* __restrict__ bloom,
    uint32_t bloom_wbits, uint32_t n_tiles)
{
    using Tr = HashTraits<uint32_t>;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bloom[];
    __shared__ FtShared sh;

    const int tid = threadIdx.x;
    const int k = a.k, w = a.w;
    const uint32_t kmask = (1u << (2 * k)) - 1;
    const uint32_t kbits = (1u << k) - 1; // k consecutive base flags
    const int64_t n_bases = (int64_t)a.n_bases;
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const uint32_t* __restrict__ slot_key = reinterpret_cast<const uint32_t*>(a.slot_key);
    const int base0 = tid * FT_G;

    for (uint32_t i = tid; i < (1u << bloom_wbits); i += FT_THREADS) s_bloom[i] = bloom[i];

    auto load_tile = [&](uint32_t tile, uint4& main, uint4& extra) {
        const int64_t origin = (int64_t)tile * FT_EVAL - FT_HALO;
        auto ld = [&](int v) -> uint4 {
            const int64_t g = origin + (int64_t)v * 16;
            if (g >= 0 && g + 16 <= n_bases) return *reinterpret_cast<const uint4*>(a.bases + g);
            uint32_t t4[4];
            for (int q = 0; q < 4; ++q) {
                uint32_t wd = 0;
                for (int b = 0; b < 4; ++b) {
                    const int64_t gg = g + q * 4 + b;
                    wd |= (uint32_t)((gg >= 0 && gg < n_bases) ? a.bases[gg] : (uint8_t)'N') << (8 * b);
                }
                t4[q] = wd;
            }
            return make_uint4(t4[0], t4[1], t4[2], t4[3]);
        };
        main = ld(tid);
        if (tid < FT_WORDS - FT_THREADS) extra = ld(FT_THREADS + tid);
    };

    // k-mer at tile position p: g = canonical hash + 1, or 0 if it holds an N or straddles two reads
    auto kmer_at = [&](int p, uint32_t& f_out, bool& strand) -> uint32_t {
        const int v = p >> 4, o = p & 15;
        const uint32_t f = __funnelshift_l(sh.pack[v + 1], sh.pack[v], 2 * o) >> (32 - 2 * k);
        f_out = f;
        const uint32_t nm = (uint32_t)sh.nmask[v] | ((uint32_t)sh.nmask[v + 1] << 16);
        if ((nm >> o) & kbits) return 0;
        const int q = p + 1; // read starts at p+1 .. p+k-1 split the k-mer
        const uint32_t st = __funnelshift_r(sh.start[q >> 5], sh.start[(q >> 5) + 1], q & 31);
        if (st & (kbits >> 1)) return 0;
        const uint32_t hf = Tr::mix(f, kmask), hr = Tr::mix(revcomp_code(f, k), kmask);
        strand = hf <= hr;
        return (hf < hr ? hf : hr) + 1;
    };

    uint4 cur = make_uint4(0, 0, 0, 0), cur_x = cur, nxt = cur, nxt_x = cur;
    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile, cur, cur_x);
    __syncthreads(); // bloom filter in place

    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t origin = (int64_t)tile * FT_EVAL - FT_HALO;
        const uint32_t first_read = a.tile_first_read[tile];
        const uint32_t next_tile = tile + gridDim.x;

        // ---- pack this tile into LDS ----
        {
            uint32_t pk, nm;
            pack16(cur, pk, nm);
            sh.pack[tid] = pk;
            sh.nmask[tid] = (uint16_t)nm;
            if (tid < FT_WORDS - FT_THREADS) {
                pack16(cur_x, pk, nm);
                sh.pack[FT_THREADS + tid] = pk;
                sh.nmask[FT_THREADS + tid] = (uint16_t)nm;
            }
            for (int i = tid; i < (FT_CODES + 31) / 32 + 1; i += FT_THREADS) sh.start[i] = 0;
            if (tid == 0) {
                sh.pack[FT_WORDS] = 0;
                sh.nmask[FT_WORDS] = 0xFFFF;
            }
        }
        __syncthreads();
        {
            const int64_t end_pos = origin + FT_CODES;
            for (uint32_t r = first_read + tid; r < a.n_reads; r += FT_THREADS) {
                const int64_t o = (int64_t)a.offsets[r];
                if (o >= end_pos) break;
                const int oc = (int)(o - origin);
                atomicOr(&sh.start[oc >> 5], 1u << (oc & 31));
            }
        }
        // prefetch: the next tile's bases stay in flight under the Bloom phase (issued after this tile's own
        // offset loads, because vector-memory results return in issue order)
        if (next_tile < n_tiles) load_tile(next_tile, nxt, nxt_x);
        // ---- Bloom test of my 16 positions ----
        uint32_t cand = 0;
        if (base0 >= FT_HALO && base0 < FT_NPOS - FT_HALO) {
            const uint32_t w0 = sh.pack[tid], w1 = sh.pack[tid + 1];
            const int sh_k = 32 - 2 * k, sh_w = 32 - (int)bloom_wbits;
#pragma unroll
            for (int j = 0; j < FT_G; ++j) {
                const uint32_t f = __funnelshift_l(w1, w0, 2 * j) >> sh_k;
                const uint32_t hsh = f * 0x9E3779B1u;
                const uint32_t word = s_bloom[hsh >> sh_w];
                cand |= ((word >> (hsh & 31)) & (word >> ((hsh >> 5) & 31)) & 1u) << j;
            }
        }
        // ---- rounds: compact candidates, exact lookup, window test, emit ----
        while (true) {
            if (tid == 0) {
                sh.ncand = 0;
                sh.nhit = 0;
            }
            __syncthreads(); // also orders the start bitmap before its first use
            if (cand) {
                const int j = __ffs(cand) - 1;
                cand &= cand - 1;
                sh.cand[atomicAdd(&sh.ncand, 1u)] = (uint16_t)(base0 + j);
            }
            const int more = __syncthreads_or(cand != 0);
            const uint32_t ncand = sh.ncand;
            // exact: is the candidate's canonical hash an index key?
            if ((uint32_t)tid < ncand) {
                const int p = sh.cand[tid];
                uint32_t f;
                bool strand = false;
                const uint32_t g = kmer_at(p, f, strand);
                if (g) {
                    const uint32_t h = g - 1;
                    uint32_t s = table_slot_dev((uint64_t)h, a.table_bits);
                    bool found = false;
                    while (true) {
                        const uint32_t key = slot_key[s];
                        if (key == h) { found = true; break; }
                        if (key == Tr::EMPTY) break;
                        s = (s + 1) & tmask;
                    }
                    if (found) {
                        const uint32_t at = atomicAdd(&sh.nhit, 1u);
                        sh.hit_slot[at] = s;
                        sh.hit_pos[at] = (uint16_t)(p | (strand ? 0x8000 : 0));
                    }
                }
            }
            __syncthreads();
            const uint32_t nhit = sh.nhit;
            const int span = 2 * w - 1;
            for (uint32_t c0 = 0; c0 < nhit; c0 += FT_VER_HITS) {
                const uint32_t nchunk = nhit - c0 < (uint32_t)FT_VER_HITS ? nhit - c0 : (uint32_t)FT_VER_HITS;
                for (uint32_t t = tid; t < nchunk * (uint32_t)span; t += FT_THREADS) {
                    const uint32_t hi = t / (uint32_t)span;
                    const int d = (int)(t - hi * span);
                    const int p = (int)(sh.hit_pos[c0 + hi] & 0x7FFF) + d - (w - 1);
                    uint32_t f;
                    bool st;
                    sh.ver[hi][d] = kmer_at(p, f, st);
                }
                __syncthreads();
                if ((uint32_t)tid < nchunk) {
                    const uint32_t* v = sh.ver[tid];
                    const uint32_t g = v[w - 1];
                    int got = 0;
                    const int need = w - 1;
                    for (int d = 1; d <= need; ++d) { // neighbours >= g on the left ...
                        const uint32_t x = v[w - 1 - d];
                        if (x == 0 || x < g) break;
                        ++got;
                    }
                    for (int d = 1; got < need && d <= need; ++d) { // ... and on the right
                        const uint32_t x = v[w - 1 + d];
                        if (x == 0 || x < g) break;
                        ++got;
                    }
                    if (got >= need) { // a window of w valid k-mers around p has no smaller hash: p is a minimizer
                        const int p = sh.hit_pos[c0 + tid] & 0x7FFF;
                        const uint32_t strand = sh.hit_pos[c0 + tid] >> 15;
                        const uint2 rec = a.slot_rec[sh.hit_slot[c0 + tid]];
                        const uint64_t gp = (uint64_t)(origin + p);
                        const uint32_t read = find_read_from(a.offsets, a.n_reads, first_read ? first_read - 1 : 0, gp);
                        const uint64_t pos = gp - a.offsets[read];
                        const unsigned long long at = atomicAdd(a.n_hits, (unsigned long long)rec.y);
                        atomicAdd(a.n_minimizers, 1ull);
                        if (at + rec.y > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
                            atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
                        } else {
                            for (uint32_t q = 0; q < rec.y; ++q) {
                                const uint32_t kn = a.rec_knode[rec.x + q];
                                const uint32_t prg = a.rec_prg[rec.x + q];
                                const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                                a.hit_key[at + q] = pack_hit_key(read, prg, rev, (uint32_t)pos);
                                a.hit_val[at + q] = kn >> 1;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            if (!more) break;
        }
        __syncthreads(); // everyone is done with this tile's LDS before it is overwritten
        cur = nxt;
        cur_x = nxt_x;
    }
}

// ---------------------------------------------------------------------------------------------
// K3: clustering on the sorted hit list
// ---------------------------------------------------------------------------------------------
// a hit opens a new cluster when read / prg / strand change or the read-position gap exceeds max_diff
__global__ void cluster_flag_kernel(const uint64_t* __restrict__ key, uint32_t n, int max_diff, uint32_t* __restrict__ head)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f = 1;
    if (i > 0) {
        uint64_t a = key[i - 1], b = key[i];
        bool same_group = (a >> HIT_POS_BITS) == (b >> HIT_POS_BITS);
        int64_t gap = (int64_t)(b & HIT_POS_MASK) - (int64_t)(a & HIT_POS_MASK);
        f = (!same_group || gap > (int64_t)max_diff) ? 1u : 0u;
    }
    head[i] = f;
}

// cid[i] = inclusive scan of head - 1; heads write their index into cstart[cid]
__global__ void cluster_start_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ scan, uint32_t n,
    uint32_t* __restrict__ cstart)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (head[i]) cstart[scan[i] - 1] = i;
    if (i == n - 1) cstart[scan[i]] = n; // sentinel
}

// size threshold of pandora define_clusters
__global__ void cluster_eval_kernel(ClusterArgs a)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= *a.d_n_clusters) return;
    uint32_t s = a.cstart[c], e = a.cstart[c + 1];
    uint64_t k0 = a.key[s], k1 = a.key[e - 1];
    uint32_t read = hit_read(k0), prg = hit_prg(k0);
    uint64_t len = a.offsets[read + 1] - a.offsets[read];
    uint64_t expected = len * 2 / (uint64_t)(a.w + 1);
    uint64_t m = a.prg_min_path_len[prg];
    if (expected < m) m = expected;
    uint32_t length_based = (uint32_t)((double)m * a.fraction);
    uint32_t thr = length_based > a.min_cluster_size ? length_based : a.min_cluster_size;
    uint32_t n = e - s;
    ClusterRec r;
    r.read = read;
    r.prg_rev = (prg << 1) | hit_rev(k0);
    r.first_pos = (uint32_t)(k0 & HIT_POS_MASK);
    r.last_pos = (uint32_t)(k1 & HIT_POS_MASK);
    r.n = n;
    r.state = n > thr ? 1u : 0u; // 1 = kept by the size filter
    a.clusters[c] = r;
}

// pandora clusterComp: first hit position, larger first, prg, forward first
__device__ inline bool cluster_before(const ClusterRec& x, const ClusterRec& y)
{
    if (x.first_pos != y.first_pos) return x.first_pos < y.first_pos;
    if (x.n != y.n) return x.n > y.n;
    return x.prg_rev < y.prg_rev; // prg, then rev=0 (forward) first
}

// pandora filter_clusters: one thread per read sweeps the read's kept clusters in cluster order
__global__ void cluster_filter_kernel(ClusterArgs a)
{
    uint32_t c0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_clusters = *a.d_n_clusters;
    if (c0 >= n_clusters) return;
    if (c0 > 0 && a.clusters[c0 - 1].read == a.clusters[c0].read) return; // not the first cluster of its read
    uint32_t read = a.clusters[c0].read;
    // gather kept clusters of this read into order[c0..), insertion-sorted
    uint32_t m = 0;
    for (uint32_t c = c0; c < n_clusters && a.clusters[c].read == read; ++c) {
        if (a.clusters[c].state == 0) continue;
        uint32_t p = m++;
        while (p > 0 && cluster_before(a.clusters[c], a.clusters[a.order[c0 + p - 1]])) {
            a.order[c0 + p] = a.order[c0 + p - 1];
            --p;
        }
        a.order[c0 + p] = c;
    }
    if (m == 0) return;
    uint32_t prev = a.order[c0];
    for (uint32_t q = 1; q < m; ++q) {
        uint32_t cur = a.order[c0 + q];
        const ClusterRec& P = a.clusters[prev];
        const ClusterRec& C = a.clusters[cur];
        bool same_prg_other_strand = (P.prg_rev >> 1) == (C.prg_rev >> 1) && (P.prg_rev & 1) != (C.prg_rev & 1);
        bool contained = C.last_pos <= P.last_pos;
        if (same_prg_other_strand || contained) {
            if (P.n >= C.n) {
                a.clusters[cur].state = 0;
            } else {
                a.clusters[prev].state = 0;
                prev = cur;
            }
        } else {
            prev = cur;
        }
    }
}

// per surviving cluster: pangraph node read count; per hit of a surviving cluster: coverage += 1
__global__ void cluster_count_kernel(ClusterArgs a)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= *a.d_n_clusters) return;
    if (a.clusters[c].state == 0) return;
    atomicAdd(&a.prg_reads[a.clusters[c].prg_rev >> 1], 1u);
    atomicAdd(a.n_clusters_kept, 1ull);
    atomicAdd(a.n_hits_kept, (unsigned long long)a.clusters[c].n);
}

__global__ void accumulate_kernel(ClusterArgs a, uint32_t n_hits)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    uint32_t c = a.scan[i] - 1;
    if (a.clusters[c].state == 0) return;
    uint32_t rev = hit_rev(a.key[i]);
    atomicAdd(&a.covg[2 * (size_t)a.val[i] + rev], 1u);
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
#define HIP_TRY(x)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return e_;                                                             \
    } while (0)

uint32_t sketch_tile_eval(int halo) { return (uint32_t)(SK_NPOS - 2 * halo); }

uint32_t sketch_n_tiles(uint64_t n_bases, int halo)
{
    uint64_t t_eval = sketch_tile_eval(halo);
    return (uint32_t)((n_bases + t_eval - 1) / t_eval);
}

hipError_t launch_sketch_probe(const SketchArgs& a, bool wide_hash, hipStream_t stream)
{
    if (a.n_bases == 0) return hipSuccess;
    const uint32_t grid = sketch_n_tiles(a.n_bases, a.halo); // positions past n_bases-k are invalid inside the kernel
    hipLaunchKernelGGL(tile_first_read_kernel, dim3((grid + 255) / 256), dim3(256), 0, stream, a.offsets, a.n_reads,
        (int)sketch_tile_eval(a.halo), a.halo, grid, a.tile_first_read);
    HIP_TRY(hipGetLastError());
    const dim3 g(grid), b(SK_THREADS);
    if (wide_hash)
        hipLaunchKernelGGL((sketch_probe_kernel<uint64_t, 0, 0>), g, b, 0, stream, a);
    else if (a.k == 15 && a.w == 11)
        hipLaunchKernelGGL((sketch_probe_kernel<uint32_t, 15, 11>), g, b, 0, stream, a);
    else if (a.k == 15 && a.w == 14)
        hipLaunchKernelGGL((sketch_probe_kernel<uint32_t, 15, 14>), g, b, 0, stream, a);
    else
        hipLaunchKernelGGL((sketch_probe_kernel<uint32_t, 0, 0>), g, b, 0, stream, a);
    return hipGetLastError();
}

uint32_t filter_n_tiles(uint64_t n_bases) { return (uint32_t)((n_bases + FT_EVAL - 1) / FT_EVAL); }

hipError_t launch_sketch_filter(const SketchArgs& a, const uint32_t* bloom, uint32_t bloom_wbits, int n_cus, hipStream_t stream)
{
    if (a.n_bases == 0) return hipSuccess;
    const uint32_t n_tiles = filter_n_tiles(a.n_bases);
    hipLaunchKernelGGL(tile_first_read_ft_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, a.offsets, a.n_reads,
        n_tiles, a.tile_first_read);
    HIP_TRY(hipGetLastError());
    const size_t dyn = sizeof(uint32_t) << bloom_wbits;
    static size_t configured = 0;
    if (dyn > configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_filter_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        configured = dyn;
    }
    // persistent grid: as many workgroups as stay resident (LDS-limited), never more than there are tiles
    const size_t lds_per_wg = dyn + sizeof(FtShared) + 64;
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds_per_wg);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    uint32_t grid = (uint32_t)n_cus * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(sketch_filter_kernel, dim3(grid), dim3(FT_THREADS), dyn, stream, a, bloom, bloom_wbits, n_tiles);
    return hipGetLastError();
}

size_t sort_temp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
        (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    return bytes;
}

size_t scan_temp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}

hipError_t sort_hits(void* temp, size_t temp_bytes, const uint64_t* key_in, uint64_t* key_out, const uint32_t* val_in,
    uint32_t* val_out, uint32_t n, hipStream_t stream)
{
    return rocprim::radix_sort_pairs(temp, temp_bytes, key_in, key_out, val_in, val_out, n, 0, 64, stream);
}

hipError_t launch_cluster_flags(const uint64_t* key, uint32_t n, int max_diff, uint32_t* head, uint32_t* scan, void* temp,
    size_t temp_bytes, hipStream_t stream)
{
    const int B = 256;
    hipLaunchKernelGGL(cluster_flag_kernel, dim3((n + B - 1) / B), dim3(B), 0, stream, key, n, max_diff, head);
    HIP_TRY(hipGetLastError());
    return rocprim::inclusive_scan(temp, temp_bytes, head, scan, n, rocprim::plus<uint32_t>(), stream);
}

hipError_t launch_cluster_starts(const uint32_t* head, const uint32_t* scan, uint32_t n, uint32_t* cstart, hipStream_t stream)
{
    const int B = 256;
    hipLaunchKernelGGL(cluster_start_kernel, dim3((n + B - 1) / B), dim3(B), 0, stream, head, scan, n, cstart);
    return hipGetLastError();
}

hipError_t launch_cluster_pipeline(const ClusterArgs& a, uint32_t n_hits, hipStream_t stream)
{
    const int B = 128;
    dim3 gc((n_hits + B - 1) / B); // n_clusters <= n_hits; the true count is read on the device
    hipLaunchKernelGGL(cluster_eval_kernel, gc, dim3(B), 0, stream, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(cluster_filter_kernel, gc, dim3(B), 0, stream, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(cluster_count_kernel, gc, dim3(B), 0, stream, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(accumulate_kernel, dim3((n_hits + 255) / 256), dim3(256), 0, stream, a, n_hits);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
