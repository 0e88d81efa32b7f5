// sketch_block.h -- the register-resident (w,k)-minimizer block shared by sketch_wave_kernel (sketch_wave.hip: every k-mer of a
// batch) and read_verify_kernel (read_verify.hip: every k-mer of the reads that hold index k-mers).  A lane owns 16 consecutive k-mer
// start positions = 16 bases, packed to 2 bits twice (first base in the low bits: `le`; first base in the high bits: `be`); the right
// neighbour's two words arrive by a DPP wave shift; k-mer j is one v_alignbit_b32 out of each stream; the hash is mix_k<K>
// (device_common.h); the W-1 hash values either side of a lane's 16 come from the neighbouring lanes by DPP wave shifts and the
// window minima are three-input minima in registers.  It restates, inside the external `pandora map` process that
// /root/reference/src/lib.rs:580-642 spawns, Seq::minimizer_sketch.
#pragma once
#include "device_common.h"

namespace drprg {
namespace dev {

constexpr int SB_G = 16; // k-mer positions (= bases) per lane

__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) // lane i <- lane i + 1 (lane 63 <- 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v) // lane i <- lane i - 1 (lane 0 <- 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// four ASCII bases -> selector bytes (A0 C1 T2 G3 = bits 2:1 of the letter); *diff receives a non-zero byte for every base
// that is not ACGT / acgt
__device__ __forceinline__ uint32_t select4(uint32_t word, uint32_t& diff_acc)
{
    const uint32_t sel = (word >> 1) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, sel); // the upper-case letter each selector stands for
    diff_acc |= (word & 0xDFDFDFDFu) ^ expect;
    return sel;
}
// bit i set iff byte i of x is non-zero
__device__ __forceinline__ uint32_t nonzero_bytes4(uint32_t x)
{
    const uint32_t t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    return ((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 0xFu;
}

// 16 ASCII bases -> the two 2-bit streams in the letter code A0 C1 T2 G3 (sketch_letters_to_hash_order turns them into the hash's
// A0 C1 G2 T3); diff != 0 iff one of the bases is not ACGT / acgt (sketch_bad16 then says which)
__device__ __forceinline__ void sketch_pack_ascii(const uint4& in, uint32_t& le, uint32_t& be, uint32_t& diff)
{
    const uint32_t s0 = select4(in.x, diff), s1 = select4(in.y, diff), s2 = select4(in.z, diff), s3 = select4(in.w, diff);
    // v_dot4_u32_u8 packs four selectors into a byte: weights 1,4,16,64 (first base lowest) / 64,16,4,1 (first base highest)
    le = __builtin_amdgcn_udot4(s0, 0x40100401u, 0u, false) | (__builtin_amdgcn_udot4(s1, 0x40100401u, 0u, false) << 8)
        | (__builtin_amdgcn_udot4(s2, 0x40100401u, 0u, false) << 16) | (__builtin_amdgcn_udot4(s3, 0x40100401u, 0u, false) << 24);
    be = (__builtin_amdgcn_udot4(s0, 0x01041040u, 0u, false) << 24) | (__builtin_amdgcn_udot4(s1, 0x01041040u, 0u, false) << 16)
        | (__builtin_amdgcn_udot4(s2, 0x01041040u, 0u, false) << 8) | __builtin_amdgcn_udot4(s3, 0x01041040u, 0u, false);
}
// one word of a 2-bit packed batch (16 letters, first base lowest) -> the two streams
__device__ __forceinline__ void sketch_pack_word(uint32_t wd, uint32_t& le, uint32_t& be)
{
    const uint32_t r = __brev(wd);
    le = wd;
    be = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
__device__ __forceinline__ uint32_t sketch_letters_to_hash_order(uint32_t x) { return x ^ ((x >> 1) & 0x55555555u); } // A0 C1 T2 G3 -> A0 C1 G2 T3
// bit i: base i of the 16 is not ACGT / acgt (only called when sketch_pack_ascii reported a difference)
__device__ __forceinline__ uint32_t sketch_bad16(const uint4& in)
{
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    (void)select4(in.x, d0); (void)select4(in.y, d1); (void)select4(in.z, d2); (void)select4(in.w, d3);
    return nonzero_bytes4(d0) | (nonzero_bytes4(d1) << 4) | (nonzero_bytes4(d2) << 8) | (nonzero_bytes4(d3) << 12);
}

// canonical hash + 1 of my 16 k-mers (0 = invalid: bit j of validbits clear) and bit j of strandbits = the forward k-mer of
// position j is the canonical one.  le / be: my words (hash order), le_next / be_next: the right neighbour's.
template <int K>
__device__ __forceinline__ void sketch_hashes16(uint32_t le, uint32_t be, uint32_t le_next, uint32_t be_next, uint32_t validbits, uint32_t (&hv)[SB_G],
    uint32_t& strandbits)
{
    strandbits = 0;
#pragma unroll
    for (int j = 0; j < SB_G; ++j) {
        // low-first stream: bits [2j, 2j + 2K) = the k-mer read backwards; its complement is the reverse-complement k-mer
        const uint32_t e = __builtin_amdgcn_alignbit(le_next, le, 2 * j);
        // high-first stream (be : be_next): the forward k-mer sits at bits [64 - 2j - 2K, 64 - 2j)
        constexpr int TOP = 64 - 2 * K;
        const int sh = TOP - 2 * j;
        const uint32_t f = sh >= 32 ? be >> (sh - 32) : __builtin_amdgcn_alignbit(be, be_next, sh);
        const uint32_t hf = mix_k<K>(f), hr = mix_k<K>(~e);
        strandbits |= (uint32_t)(hf <= hr) << j;
        const uint32_t h1 = (hf < hr ? hf : hr) + 1u;
        hv[j] = h1 & (uint32_t)__builtin_amdgcn_sbfe((int)validbits, j, 1);
    }
}

// window minimizers: bit j of the result = position j is one, i.e. some window of W consecutive valid k-mers containing it has no
// smaller value (an invalid k-mer is 0: a window holding one has minimum 0, which no valid value equals).  The W-1 values either
// side come from the neighbouring lanes (lane 0's left and lane 63's right neighbour read as invalid / as lane 0).  The caller
// masks the result with validbits.
template <int W> __device__ __forceinline__ uint32_t sketch_minimizers16(const uint32_t (&hv)[SB_G])
{
    static_assert(W >= 2 && W - 1 <= SB_G, "the neighbours of a lane's positions lie in the two adjacent lanes");
    uint32_t minbits = 0;
    constexpr int N = SB_G + 2 * (W - 1);
    uint32_t g[N];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) g[i] = from_prev_lane(hv[SB_G - (W - 1) + i]);
#pragma unroll
    for (int j = 0; j < SB_G; ++j) g[W - 1 + j] = hv[j];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) g[W - 1 + SB_G + i] = from_next_lane(hv[i]);
    constexpr int NW = SB_G + W - 1; // window starts that matter: 0 .. NW - 1
    uint32_t wm[NW];                  // wm[i] = min g[i .. i + W - 1]
    if constexpr (W == 11 || W == 14) {
        // three-input minima: runs of 3, of 9, then the window (11 = 9 + a run of 3 that overlaps it, 14 = 9 + two runs of 3)
        uint32_t m3[N - 2], m9[N - 8];
#pragma unroll
        for (int i = 0; i < N - 2; ++i) m3[i] = min(min(g[i], g[i + 1]), g[i + 2]);
#pragma unroll
        for (int i = 0; i < N - 8; ++i) m9[i] = min(min(m3[i], m3[i + 3]), m3[i + 6]);
#pragma unroll
        for (int i = 0; i < NW; ++i) wm[i] = W == 11 ? min(m9[i], m3[i + 8]) : min(min(m9[i], m3[i + 9]), m3[i + 11]);
        // the same shape with maxima over the W windows that hold position j: windows j .. j + W - 1
        uint32_t x3[NW - 2], x9[NW - 8];
#pragma unroll
        for (int i = 0; i < NW - 2; ++i) x3[i] = max(max(wm[i], wm[i + 1]), wm[i + 2]);
#pragma unroll
        for (int i = 0; i < NW - 8; ++i) x9[i] = max(max(x3[i], x3[i + 3]), x3[i + 6]);
#pragma unroll
        for (int j = 0; j < SB_G; ++j) {
            const uint32_t best = W == 11 ? max(x9[j], x3[j + 8]) : max(max(x9[j], x3[j + 9]), x3[j + 11]);
            minbits |= (uint32_t)(best == hv[j]) << j;
        }
    } else {
        constexpr int P = (W >= 16) ? 16 : (W >= 8) ? 8 : (W >= 4) ? 4 : 2; // largest power of two <= W
#pragma unroll
        for (int sp = 1; sp < P; sp *= 2) {
#pragma unroll
            for (int i = 0; i + sp < N; ++i) g[i] = g[i] < g[i + sp] ? g[i] : g[i + sp];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) wm[i] = g[i] < g[i + W - P] ? g[i] : g[i + W - P];
#pragma unroll
        for (int sp = 1; sp < P; sp *= 2) {
#pragma unroll
            for (int i = 0; i + sp < NW; ++i) wm[i] = wm[i] > wm[i + sp] ? wm[i] : wm[i + sp];
        }
#pragma unroll
        for (int j = 0; j < SB_G; ++j) {
            const uint32_t best = wm[j] > wm[j + W - P] ? wm[j] : wm[j + W - P];
            minbits |= (uint32_t)(best == hv[j]) << j;
        }
    }
    return minbits;
}

} // namespace dev
} // namespace drprg
