// sketch_filter.hip -- K1+K2 in their fast form: persistent waves + an LDS-resident Bloom prefilter, and hits that
// leave the pipeline already ordered by (read, position).
//
// A read minimizer can only produce a hit if its k-mer is an index k-mer (in either orientation).  So instead of
// hashing every k-mer of every read (sketch_probe.hip: ~86 VALU instructions per base, two 15-op hashes each):
//
//   sketch_filter_kernel   every wave streams one contiguous chunk of the concatenated base buffer, packs it to 2 bits
//                          per base in registers and tests each position's k-mer *code* against a Bloom filter of the
//                          index k-mer codes that stays in LDS for the lifetime of the workgroup; the positions that
//                          pass (index k-mers + ~0.02 % false positives) are appended in position order to the wave's
//                          own slice of a candidate buffer (cursor in a scalar register: no atomics, no barriers).
//                          Three tiles of bases are in flight per wave in a register ring, the right neighbour's
//                          packed word arrives through a DPP wave shift.
//   cand_scan_kernel       one workgroup: exclusive scan of the slice counts; cand_gather_kernel copies the slices into
//                          one dense, ordered candidate list.
//   verify_count_kernel    one lane per candidate, start to finish, no barrier and no atomic: canonical hash from the
//                          raw bases -> exact table lookup (false positives end here) -> read lookup -> window-minimizer
//                          test over the 2w-1 neighbouring k-mers inside the read, hashed one by one out of a register
//                          shift register.  Leaves (slot, read, strand, pos) per candidate and the hit count per
//                          workgroup.
//   hit_scan_kernel        one workgroup: exclusive scan of the workgroup hit counts, batch totals.
//   expand_kernel          one (key,val) hit per index record of every minimizer, at its scanned offset: the hit list
//                          is ordered by (read, position), which leaves only a tiny per-read reorder by (prg, strand)
//                          (read_sort_kernel) instead of a global 64-bit radix sort when the reads are short.
//
// The Bloom filter has no false negatives and every survivor is re-derived exactly from the bases, so the result is
// identical to the direct kernel (tests/test_gpu_parity.py checks both against the oracle).  Serves k <= 15, w <= 16
// and indexes whose filter fits 64 KB of LDS; everything else takes the direct kernel.
#include "common.h"
#include "device_common.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace drprg {
namespace dev {

constexpr int FT_THREADS = 1024;
constexpr int FT_WAVES = FT_THREADS / 64;
constexpr int FT_G = 32;                // positions per lane
constexpr int FT_WPOS = 63 * FT_G;      // positions per wave tile: lane 63's word is only lane 62's right neighbour
constexpr int FT_BLOOM_WORDS = 1 << 14; // levels 1+2 of the filter, at most (64 KB of LDS)
constexpr int FT_L0_WORDS = 1 << 15;    // level 0 (128 KB; levels 1+2 then get 32 KB: all 160 KB of a CU)
constexpr int EX_THREADS = 256;
constexpr int SCAN_THREADS = 1024;
constexpr int FT_SUB = 2;                    // slices per filter wave
constexpr int MAX_SLICES = SCAN_THREADS * 8;
constexpr int MAX_EX_WG = SCAN_THREADS * 4;  // workgroups of verify_count_kernel / expand_kernel

// 16 ASCII bases -> 32 bits, 2 per base, first base in the lowest bits.  The 2-bit letter is bits 2:1 of the ASCII code
// (A 0, C 1, T 2, G 3, either case; anything else aliases one of them: it can only create a false candidate, which
// verify_count_kernel rejects from the raw bases).  One multiply gathers the four fields of a dword into its top byte.
__device__ inline uint32_t pack16le(const uint4& in)
{
    constexpr uint32_t M = (1u << 23) | (1u << 17) | (1u << 11) | (1u << 5);
    const uint32_t p0 = (in.x & 0x06060606u) * M, p1 = (in.y & 0x06060606u) * M;
    const uint32_t p2 = (in.z & 0x06060606u) * M, p3 = (in.w & 0x06060606u) * M;
    // byte 3 of p0..p3 -> bytes 0..3
    return __builtin_amdgcn_perm(p1, p0, 0x0c0c0703u) | __builtin_amdgcn_perm(p3, p2, 0x07030c0cu);
}


// Bloom filter layout (built by FlatIndex, index.cpp; both orientations of every index k-mer are entered):
//   code   = little-endian 2-bit letters of the k-mer (pack16le alphabet), 2k bits
//   level 1, keyed on the first min(k,12) bases: x = code & 0xFFFFFF, h = (x * BLOOM_C1) mod 2^32,
//            word (h >> 18) & (words-1), bits 31-(h & 31), 31-((h >> 8) & 31), 31-((x >> 16) & 31)
//   level 2, keyed on the whole code: h2 = code * BLOOM_C2, word h2 >> (32-wbits), bits h2 & 31, (h2>>5) & 31, (h2>>10) & 31
//   level 0 (own array, k = 15 and small indexes), keyed on the 12-mers at offsets 0..3 of the k-mer: like level 1 with
//            BLOOM_C0; probed once per four read positions
// A level-0/1 probe costs 7 VALU + 1 LDS instruction (24-bit multiply, byte-select shifts, 3-input AND, funnel-shift
// accumulate).  Without level 0 every position pays one; with it every fourth does, level 1 runs for the ~5 % of the
// groups that pass, level 2 for the ~1 % level-1 survivors.
//
// The filter lives in dynamic LDS, which starts at LDS address 0 (the kernel has no static LDS): a masked hash is the
// address of its word, read through an address-space-3 pointer made from the integer.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ uint32_t lds_at(uint32_t byte_addr) { return *(lds_cu32*)(uintptr_t)byte_addr; }
// bit 31 of the result: all three filter bits of (h, x) are set in word
__device__ __forceinline__ uint32_t bloom_test(uint32_t word, uint32_t h, uint32_t x)
{
    uint32_t t2; // word << h[12:8]: the byte select is free, the compiler does not find it
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(t2) : "v"(h), "v"(word));
    return (word << (h & 31)) & t2 & (word << ((x >> 16) & 0xFF));
}

// SHORT_K: k < 12, the level-1 key must be masked to 2k bits.  LEVEL0: the level-0 array is present (k = 15).
template <bool SHORT_K, bool LEVEL0>
__global__ __launch_bounds__(FT_THREADS) void sketch_filter_kernel(SketchArgs a, FilterWork fw)
{
    extern __shared__ uint32_t s_dyn[]; // [level 0: FT_L0_WORDS] [levels 1+2: 2^bloom_wbits words]
    constexpr uint32_t L12_BASE = LEVEL0 ? FT_L0_WORDS * 4u : 0u;

    const int tid = threadIdx.x, lane = tid & 63;
    const int k = a.k;
    const int64_t n_bases = (int64_t)a.n_bases;
    const uint32_t n_words = 1u << fw.bloom_wbits;
    const uint32_t amask = (n_words - 1) << 2; // byte offset of the level-1 word = (h >> 16) & amask
    const uint32_t amask0 = ((1u << fw.bloom0_wbits) - 1) << 2;
    const uint32_t kmask = (k < 16) ? ((1u << (2 * k)) - 1) : 0xFFFFFFFFu;
    const uint32_t kmask24 = kmask & 0xFFFFFFu;
    const int sh_w = 32 - (int)fw.bloom_wbits;

    if (LEVEL0)
        for (uint32_t i = tid; i < (1u << fw.bloom0_wbits); i += FT_THREADS) s_dyn[i] = fw.bloom0[i];
    if (!LEVEL0) // (the level-0 form leaves levels 1+2 to refine_kernel)
        for (uint32_t i = tid; i < n_words; i += FT_THREADS) s_dyn[L12_BASE / 4 + i] = fw.bloom[i];
    if (tid == 0 && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_dyn != 0u) atomicOr(a.overflow, 8u);

    // every wave owns a contiguous range of tiles and FT_SUB consecutive slices of the candidate buffers: it moves on to its
    // next slice every tiles_per_slice tiles (more, shorter slices: more parallelism for refine_kernel / cand_gather_kernel)
    const uint32_t gw = blockIdx.x * FT_WAVES + (uint32_t)(tid >> 6);
    uint32_t tile = gw * fw.tiles_per_wave;
    const uint32_t tile_end = tile + fw.tiles_per_wave < fw.n_tiles ? tile + fw.tiles_per_wave : fw.n_tiles;
    constexpr uint32_t step = 1;
    uint32_t slice = gw * FT_SUB, next_slice_at = tile + fw.tiles_per_slice;

    auto load16 = [&](int64_t g) -> uint4 { // 16 bases at global position g (a multiple of 16)
        if (g + 16 <= n_bases) return *reinterpret_cast<const uint4*>(a.bases + g);
        uint32_t t4[4];
        for (int q = 0; q < 4; ++q) {
            uint32_t wd = 0;
            for (int b = 0; b < 4; ++b) {
                const int64_t gg = g + q * 4 + b;
                wd |= (uint32_t)(gg < n_bases ? a.bases[gg] : (uint8_t)'N') << (8 * b);
            }
            t4[q] = wd;
        }
        return make_uint4(t4[0], t4[1], t4[2], t4[3]);
    };
    struct Pair { // the 32 bases of one lane in one tile
        uint4 a, b;
    };
    // Tiles whose 64 x 32 bytes lie inside the buffer are loaded without any guard, so that the compiler can count the
    // loads in flight (a guarded byte path inside the loop forces s_waitcnt vmcnt(0) everywhere); the one or two
    // tiles at the very end of the buffer take the guarded path after the pipelined loop.
    const uint32_t n_full = n_bases >= 64 * FT_G ? (uint32_t)((n_bases - 64 * FT_G) / FT_WPOS) + 1 : 0u;
    const uint32_t full_end = tile_end < n_full ? tile_end : n_full;
    auto fetch = [&](uint32_t t, Pair& p) { // unconditional (a prefetch past the wave's range re-reads its last full tile)
        const uint32_t tc = t < full_end ? t : full_end - 1;
        const uint8_t* g = a.bases + (int64_t)tc * FT_WPOS + (int64_t)lane * FT_G;
        p.a = *reinterpret_cast<const uint4*>(g);
        p.b = *reinterpret_cast<const uint4*>(g + 16);
    };
    uint64_t* out = fw.raw_pos + (size_t)slice * fw.raw_slice;
    uint4* grp_out = LEVEL0 ? fw.raw_grp + (size_t)slice * fw.raw_slice : nullptr;
    (void)grp_out;
    uint32_t wcur = 0; // candidates in the current slice so far (wave-uniform)
    auto close_slice = [&]() {
        if (lane == 0) {
            (LEVEL0 ? fw.grp_count : fw.slice_count)[slice] = wcur;
            if (wcur > fw.raw_slice) atomicOr(a.overflow, 4u);
        }
    };

    // one tile: my 32 positions start in words wa, wb; wc (the first word of lane+1) completes the last k-mers
    auto process = [&](uint32_t t, const Pair& p) {
        if (t == next_slice_at) { // wave-uniform
            close_slice();
            ++slice;
            next_slice_at += fw.tiles_per_slice;
            out += fw.raw_slice;
            if (LEVEL0) grp_out += fw.raw_slice;
            wcur = 0;
        }
        const uint32_t wa = pack16le(p.a), wb = pack16le(p.b);
        const uint32_t wc = __builtin_amdgcn_update_dpp(0u, wa, 0x130 /* wave_shl:1: lane i <- lane i+1 */, 0xF, 0xF, false);
        if constexpr (LEVEL0) {
            // ---- level 0: one 12-mer per four positions (the one at 4g+3 lies inside every 15-mer starting at 4g..4g+3); group g
            // ends up in bit g of grp.  The ~2 % of the groups that pass leave the kernel as they are, with their bases: levels 1+2
            // run in refine_kernel, one lane per group (here they would run for the whole wave as often as its busiest lane
            // needs: a third of this kernel's instructions) ----
            uint32_t grp = 0, xs[FT_G / 4], hs[FT_G / 4], ws[FT_G / 4];
#pragma unroll
            for (int g = 0; g < FT_G / 4; ++g) { // all eight LDS reads in flight before the first test
                const int j = 4 * g + 3;
                const uint32_t lo = j < 16 ? wa : wb, hi = j < 16 ? wb : wc;
                xs[g] = __builtin_amdgcn_alignbit(hi, lo, 2 * (j & 15));
                hs[g] = __umul24(xs[g], BLOOM_C0);
                ws[g] = lds_at((hs[g] >> 15) & amask0);
            }
#pragma unroll
            for (int g = FT_G / 4 - 1; g >= 0; --g) grp = __builtin_amdgcn_alignbit(grp, bloom_test(ws[g], hs[g], xs[g]), 31);
            if (lane == 63 || (fw.debug & 1u)) grp = 0;
            // ---- append in (lane, group) = position order: exclusive prefix of the per-lane counts (0..8) from four ballots ----
            const uint32_t cnt = (uint32_t)__popc(grp);
            const uint64_t b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u), b3 = __ballot(cnt & 8u);
            if (b0 | b1 | b2 | b3) {
                auto below = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
                uint32_t at = wcur + below(b0) + 2u * below(b1) + 4u * below(b2) + 8u * below(b3);
                const uint64_t base = (uint64_t)t * FT_WPOS + (uint64_t)lane * FT_G;
                while (grp) {
                    const int g = __ffs(grp) - 1;
                    grp &= grp - 1;
                    const uint32_t lo = g < 4 ? wa : wb, hi = g < 4 ? wb : wc;
                    const uint32_t sh = (8u * (uint32_t)g) & 31u;
                    const uint64_t pos = base + 4u * (uint32_t)g;
                    if (at < fw.raw_slice) grp_out[at] = make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), __funnelshift_r(lo, hi, sh), hi >> sh);
                    ++at;
                }
                wcur += (uint32_t)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3));
            }
            return;
        }
        uint32_t cand = 0;
        if (!(fw.debug & 1u)) {
            uint32_t c1 = 0; // level-1 survivors, position j in bit j
            {
                // ---- level 1 over my 32 positions ----
#pragma unroll
                for (int j = FT_G - 1; j >= 0; --j) {
                    const uint32_t lo = j < 16 ? wa : wb, hi = j < 16 ? wb : wc;
                    uint32_t x = (j & 15) ? __builtin_amdgcn_alignbit(hi, lo, 2 * (j & 15)) : lo; // code in the low bits, later bases above
                    if (SHORT_K) x &= kmask24;
                    const uint32_t h = __umul24(x, BLOOM_C1);
                    c1 = __builtin_amdgcn_alignbit(c1, bloom_test(lds_at(L12_BASE + ((h >> 16) & amask)), h, x), 31); // c1 = c1 << 1 | bit
                }
                if (lane == 63) c1 = 0;
            }
            // ---- level 2, only for the survivors: three more bits in a second word, keyed on the whole code ----
            while (c1) {
                const int j = __ffs(c1) - 1;
                c1 &= c1 - 1;
                const uint32_t lo = j < 16 ? wa : wb, hi = j < 16 ? wb : wc;
                const uint32_t f = __funnelshift_r(lo, hi, 2 * (j & 15)) & kmask;
                const uint32_t h2 = f * BLOOM_C2;
                const uint32_t word = lds_at(L12_BASE + ((h2 >> sh_w) << 2));
                cand |= ((word >> (h2 & 31)) & (word >> ((h2 >> 5) & 31)) & (word >> ((h2 >> 10) & 31)) & 1u) << j;
            }
        }
        // ---- append in (lane, bit) = position order; about two candidates per tile survive ----
        uint64_t m = __ballot(cand != 0);
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cand, l);
            if (lane == l) {
                const uint64_t base = (uint64_t)t * FT_WPOS + (uint64_t)lane * FT_G;
                uint32_t cc = c, at = wcur;
                while (cc) {
                    const int j = __ffs(cc) - 1;
                    cc &= cc - 1;
                    if (at < fw.raw_slice) out[at] = base + (uint64_t)j;
                    ++at;
                }
            }
            wcur += (uint32_t)__popc(c);
        }
    };

    // register ring, unrolled so that no tile is copied between registers: the tile being processed plus two in
    // flight (2 KB each per wave, 16 waves: 64 KB outstanding per CU; a fourth register set made the kernel slower)
    Pair r0 {}, r1 {}, r2 {};
    const bool pipelined = tile < full_end; // wave-uniform
    if (pipelined) {
        fetch(tile, r0);
        fetch(tile + step, r1);
    }
    __syncthreads(); // Bloom filter in place; the only barrier
    while (tile < full_end) {
        fetch(tile + 2 * step, r2);
        process(tile, r0);
        tile += step;
        if (tile >= full_end) break;
        fetch(tile + 2 * step, r0);
        process(tile, r1);
        tile += step;
        if (tile >= full_end) break;
        fetch(tile + 2 * step, r1);
        process(tile, r2);
        tile += step;
    }
    for (; tile < tile_end; tile += step) { // the end of the buffer, guarded loads
        const int64_t g = (int64_t)tile * FT_WPOS + (int64_t)lane * FT_G;
        Pair p;
        p.a = load16(g);
        p.b = load16(g + 16);
        process(tile, p);
    }
    close_slice();
    for (++slice; slice < (gw + 1) * FT_SUB; ++slice) // slices this wave never reached (the last waves of a short batch)
        if (lane == 0) (LEVEL0 ? fw.grp_count : fw.slice_count)[slice] = 0;
}

// Second stage of the filter for the groups that passed level 0 (level-0 form of sketch_filter_kernel): one lane per
// group tests its four k-mer codes against a 64 KB LDS-resident filter (four bits per code, < 15 % full); every wave
// works through whole slices and compacts the surviving positions, in order, into the slice of raw_pos that
// cand_scan_kernel / cand_gather_kernel expect.
constexpr int RF_THREADS = 1024;
__global__ __launch_bounds__(RF_THREADS) void refine_kernel(SketchArgs a, FilterWork fw)
{
    extern __shared__ uint32_t s_bloom[]; // the second-stage filter: 2^BLOOMR_WBITS words
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t kmask = (1u << (2 * a.k)) - 1; // k = 15
    for (uint32_t i = tid; i < (1u << BLOOMR_WBITS); i += RF_THREADS) s_bloom[i] = fw.bloomr[i];
    __syncthreads(); // the only barrier: from here on every wave works through its own slices
    auto below = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    const uint32_t n_waves = gridDim.x * (RF_THREADS / 64);
    for (uint32_t s = blockIdx.x * (RF_THREADS / 64) + (uint32_t)(tid >> 6); s < fw.n_slices; s += n_waves) {
        const uint32_t n_raw = fw.grp_count[s], n = n_raw < fw.raw_slice ? n_raw : fw.raw_slice;
        const uint4* __restrict__ in = fw.raw_grp + (size_t)s * fw.raw_slice;
        uint64_t* __restrict__ out = fw.raw_pos + (size_t)s * fw.raw_slice;
        uint32_t written = 0; // wave-uniform
        uint4 nxt = (uint32_t)lane < n ? in[lane] : make_uint4(0, 0, 0, 0);
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const uint4 r = nxt;
            const uint32_t i = c0 + (uint32_t)lane;
            if (i + 64 < n) nxt = in[i + 64]; // in flight while this chunk is tested
            uint32_t cand = 0;
            if (i < n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { // four bits of one word, keyed on the whole code of the k-mer at position q
                    const uint32_t f = (q ? __funnelshift_r(r.z, r.w, 2 * q) : r.z) & kmask;
                    const uint32_t h = f * BLOOM_CR;
                    const uint32_t word = s_bloom[h >> (32 - BLOOMR_WBITS)];
                    cand |= ((word >> (h & 31)) & (word >> ((h >> 5) & 31)) & (word >> ((h >> 10) & 31)) & (word >> ((h >> 15) & 31)) & 1u) << q;
                }
            }
            // ordered append: exclusive prefix of the per-lane counts (0..4) from three ballots
            const uint32_t cnt = (uint32_t)__popc(cand);
            const uint64_t b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
            if (b0 | b1 | b2) {
                uint32_t at = written + below(b0) + 2u * below(b1) + 4u * below(b2);
                const uint64_t pos = ((uint64_t)r.y << 32) | r.x;
                while (cand) {
                    const int q = __ffs(cand) - 1;
                    cand &= cand - 1;
                    if (at < fw.raw_slice) out[at] = pos + (uint64_t)q;
                    ++at;
                }
                written += (uint32_t)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2));
            }
        }
        if (lane == 0) {
            fw.slice_count[s] = written;
            if (n_raw > fw.raw_slice || written > fw.raw_slice) atomicOr(a.overflow, 4u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// scans
// ---------------------------------------------------------------------------------------------
// one workgroup: cand_prefix = exclusive scan of min(slice_count, raw_slice)
__global__ __launch_bounds__(SCAN_THREADS) void cand_scan_kernel(FilterWork fw)
{
    __shared__ uint32_t s_w[SCAN_THREADS / 64 + 1];
    constexpr int PER = MAX_SLICES / SCAN_THREADS;
    const int tid = threadIdx.x;
    uint32_t v[PER], run = 0;
    for (int i = 0; i < PER; ++i) {
        const uint32_t s = (uint32_t)tid * PER + i;
        const uint32_t n = s < fw.n_slices ? fw.slice_count[s] : 0u;
        v[i] = run;
        run += n < fw.raw_slice ? n : fw.raw_slice;
    }
    uint32_t total;
    const uint32_t before = block_exclusive_scan<SCAN_THREADS / 64>(run, s_w, &total);
    for (int i = 0; i < PER; ++i) {
        const uint32_t s = (uint32_t)tid * PER + i;
        if (s < fw.n_slices) fw.cand_prefix[s] = before + v[i];
    }
    if (tid == 0) fw.cand_prefix[fw.n_slices] = total;
}

// ---------------------------------------------------------------------------------------------
// verification
// ---------------------------------------------------------------------------------------------
// 16 ASCII bases -> packed codes (A0 C1 G2 T3, first base highest) + 16-bit "not ACGT" mask (bit i = base i)
__device__ inline void pack16n(const uint4& in, uint32_t& packed, uint32_t& nmask)
{
    const uint32_t e0 = encode4(in.x), e1 = encode4(in.y), e2 = encode4(in.z), e3 = encode4(in.w);
    packed = ((((e0 & 0x03030303u) * 0x40100401u) >> 24) << 24) | ((((e1 & 0x03030303u) * 0x40100401u) >> 24) << 16)
        | ((((e2 & 0x03030303u) * 0x40100401u) >> 24) << 8) | (((e3 & 0x03030303u) * 0x40100401u) >> 24);
    nmask = 0;
    if ((e0 | e1 | e2 | e3) & 0x04040404u) { // rare; (flags * 0x01020408) >> 24 gathers the flag of byte i into bit i
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

// 16 bases at global position g (a multiple of 16); bytes past the end of the buffer read as 'N'
__device__ inline uint4 load16_guarded(const uint8_t* __restrict__ bases, int64_t n_bases, int64_t g)
{
    if (g + 16 <= n_bases) return *reinterpret_cast<const uint4*>(bases + g);
    uint32_t t4[4];
    for (int q = 0; q < 4; ++q) {
        uint32_t wd = 0;
        for (int b = 0; b < 4; ++b) {
            const int64_t gg = g + q * 4 + b;
            wd |= (uint32_t)(gg < n_bases ? bases[gg] : (uint8_t)'N') << (8 * b);
        }
        t4[q] = wd;
    }
    return make_uint4(t4[0], t4[1], t4[2], t4[3]);
}

// the contiguous range of the ordered candidate list that workgroup `wg` of `n_wg` owns
__device__ inline void candidate_range(const FilterWork& fw, uint32_t wg, uint32_t n_wg, uint32_t& t_begin, uint32_t& t_end)
{
    const uint32_t total = fw.cand_prefix[fw.n_slices];
    const uint32_t per_wg = (total + n_wg - 1) / n_wg;
    const uint64_t b = (uint64_t)wg * per_wg;
    t_begin = b < total ? (uint32_t)b : total;
    t_end = b + per_wg < total ? (uint32_t)(b + per_wg) : total;
}

// slices -> one dense, ordered candidate list (cand_info[t] holds the position until verify_count_kernel replaces it)
__global__ __launch_bounds__(64) void cand_gather_kernel(FilterWork fw)
{
    const uint32_t s = blockIdx.x;
    const uint32_t n = fw.cand_prefix[s + 1] - fw.cand_prefix[s];
    const uint64_t* __restrict__ src = fw.raw_pos + (size_t)s * fw.raw_slice;
    uint64_t* __restrict__ dst = fw.cand_info + fw.cand_prefix[s];
    for (uint32_t i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}

// One lane per candidate, start to finish: the 64 bases around it are packed into registers once (2 bits per base,
// first base highest), the candidate's canonical hash goes to the exact table lookup (Bloom false positives end there),
// then the 2w-1 neighbouring k-mers are hashed one after the other out of a 96-bit shift register -- a third of the
// instructions of giving every neighbour its own lane, each of which had to load and pack its own bases.  It is a read
// minimizer iff the run of neighbours with hash >= its own (inside the read, no N) reaches w-1 across both sides.
__global__ __launch_bounds__(EX_THREADS) void verify_count_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    using Tr = HashTraits<uint32_t>;
    __shared__ uint32_t s_red[3][EX_THREADS / 64];
    const int tid = threadIdx.x;
    uint32_t t_begin, t_end;
    candidate_range(fw, blockIdx.x, gridDim.x, t_begin, t_end);
    const uint32_t* __restrict__ slot_key = reinterpret_cast<const uint32_t*>(a.slot_key);
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const int k = a.k, w = a.w;
    const int sh_k = 32 - 2 * k;
    const uint32_t kmask = (1u << (2 * k)) - 1;
    const int64_t n_bases = (int64_t)a.n_bases;
    const double reads_per_base = (double)a.n_reads / (double)(a.n_bases ? a.n_bases : 1);
    uint32_t my_hits = 0, my_nmin = 0, my_maxlen = 0;
    for (uint32_t t = t_begin + tid; t < t_end; t += EX_THREADS) {
        const int64_t gp = (int64_t)fw.cand_info[t]; // position now, (slot, strand, read) when this lane is done
        uint32_t pos1 = 0, slot = 0, read = READ_NONE, strand = 0;
        uint4 crec = make_uint4(0, 0, 0, 0);
        if (gp + k <= n_bases) {
            // the read of every candidate, index k-mer or not: read_cluster_kernel finds the first candidate of a read by
            // comparing neighbours (interpolated first guess: exact for fixed-length reads, a short gallop otherwise)
            read = find_read_near(a.offsets, a.n_reads, (uint32_t)((double)gp * reads_per_base), (uint64_t)gp);
            // ---- the 64 bases [a0, a0+64) hold the candidate and all its neighbours (w <= 16, k <= 15) ----
            const int64_t a0 = (gp > 15 ? gp - 15 : 0) & ~(int64_t)15;
            uint32_t r0w, r1w, r2w, r3w, n0, n1, n2, n3;
            pack16n(load16_guarded(a.bases, n_bases, a0), r0w, n0);
            pack16n(load16_guarded(a.bases, n_bases, a0 + 16), r1w, n1);
            pack16n(load16_guarded(a.bases, n_bases, a0 + 32), r2w, n2);
            pack16n(load16_guarded(a.bases, n_bases, a0 + 48), r3w, n3);
            uint64_t bad = (uint64_t)(n0 | (n1 << 16)) | ((uint64_t)(n2 | (n3 << 16)) << 32); // bit i: base a0+i is not ACGT
            if (bad) { // -> bit i: the k-mer starting at a0+i holds such a base
                uint64_t m = bad;
                for (int i = 1; i < k; ++i) m |= bad >> i;
                bad = m;
            }
            // ---- the candidate's own canonical hash, exact lookup ----
            const int oc = (int)(gp - a0); // 0..30
            uint32_t g = 0;
            if (!((bad >> oc) & 1u)) {
                const uint32_t h0 = (oc & 16) ? r1w : r0w, h1 = (oc & 16) ? r2w : r1w;
                const uint32_t f = __funnelshift_l(h1, h0, 2 * (oc & 15)) >> sh_k;
                const uint32_t hf = Tr::mix(f, kmask), hr = Tr::mix(revcomp_code(f, k), kmask);
                strand = hf <= hr ? 1u : 0u;
                g = (hf < hr ? hf : hr) + 1;
            }
            bool found = false;
            if (g) {
                const uint32_t h = g - 1;
                uint32_t sl = table_slot_dev((uint64_t)h, a.table_bits);
                while (true) {
                    const uint32_t key = slot_key[sl];
                    if (key == h) { found = true; break; }
                    if (key == Tr::EMPTY) break;
                    sl = (sl + 1) & tmask;
                }
                slot = sl;
            }
            if (found) {
                const int64_t r0 = (int64_t)a.offsets[read], r1 = (int64_t)a.offsets[read + 1];
                if (gp + k <= r1) { // the k-mer lies inside one read
                    // ---- scan q = q_first .. q_first + 2w-2; steps outside [gp-(w-1), gp+(w-1)] or the read are invalid ----
                    const int64_t q_lo = gp - (w - 1);
                    const int64_t q_first = q_lo > a0 ? q_lo : a0; // a0 <= max(q_lo, 0)
                    const int of = (int)(q_first - a0);            // 0..30
                    const int64_t v_lo = r0 > q_first ? r0 : q_first;
                    const int64_t v_hi = (r1 - k) < (gp + w - 1) ? (r1 - k) : (gp + w - 1);
                    const int i_lo = (int)(v_lo - q_first), i_hi = (int)(v_hi - q_first), ic = (int)(gp - q_first);
                    // align the shift register on q_first: 48 bases in three words cover 2w-1 + k-1 <= 45
                    if (of & 16) { r0w = r1w; r1w = r2w; r2w = r3w; r3w = 0; }
                    const int s2 = 2 * (of & 15);
                    r0w = __funnelshift_l(r1w, r0w, s2);
                    r1w = __funnelshift_l(r2w, r1w, s2);
                    r2w = __funnelshift_l(r3w, r2w, s2);
                    // steps that can count at all: inside the read and the window, no N in the k-mer (bit i = step i)
                    const uint32_t valid = ((2u << i_hi) - 1u) & ~((1u << i_lo) - 1u) & ~(uint32_t)(bad >> of);
                    uint32_t streak = 0, right = 0, alive = 1;
                    for (int i = 0; i < 2 * w - 1; ++i) {
                        const uint32_t f = r0w >> sh_k;
                        r0w = __funnelshift_l(r1w, r0w, 2);
                        r1w = __funnelshift_l(r2w, r1w, 2);
                        r2w <<= 2;
                        const uint32_t hf = Tr::mix(f, kmask), hr = Tr::mix(revcomp_code(f, k), kmask);
                        const uint32_t x = (hf < hr ? hf : hr) + 1;
                        const bool ok = ((valid >> i) & 1u) && x >= g;
                        if (i < ic) streak = ok ? streak + 1 : 0;
                        else if (i > ic) {
                            alive = ok ? alive : 0u;
                            right += alive;
                        }
                    }
                    if ((int)(streak + right) >= w - 1) {
                        const uint64_t pos = (uint64_t)(gp - r0);
                        if (pos >= (1ull << HIT_POS_BITS)) atomicOr(a.overflow, 2u);
                        else {
                            pos1 = (uint32_t)pos + 1;
                            const uint2 rec = a.slot_rec[slot];
                            my_hits += rec.y;
                            my_nmin += 1;
                            const uint32_t len = (uint32_t)((r1 - r0) > 0xFFFFFFFFll ? 0xFFFFFFFFll : (r1 - r0));
                            my_maxlen = len > my_maxlen ? len : my_maxlen;
                            // for read_cluster_kernel: the first hit of this minimizer and the size threshold of a cluster
                            // of this read on that hit's PRG (cluster_eval_kernel)
                            const uint32_t kn = a.rec_knode[rec.x], prg = a.rec_prg[rec.x];
                            const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                            const uint64_t expected = (uint64_t)(r1 - r0) * 2 / (uint64_t)(w + 1);
                            uint64_t m = rc.prg_min_path_len[prg];
                            if (expected < m) m = expected;
                            const uint32_t length_based = (uint32_t)((double)m * rc.fraction);
                            uint32_t thr = length_based > rc.min_cluster_size ? length_based : rc.min_cluster_size;
                            if (thr > 0xFFFFu) thr = 0xFFFFu; // read_cluster_kernel stages at most RC_HCAP hits: no difference
                            crec = make_uint4(rec.x, rec.y, (strand << 31) | (((prg << 1) | rev) << 16) | thr, (kn >> 1) * 2u + rev);
                        }
                    }
                }
            }
        }
        fw.cand_pos1[t] = pos1;
        fw.cand_info[t] = ((uint64_t)slot << 32) | ((uint64_t)strand << 31) | (uint64_t)read;
        fw.cand_rec[t] = crec;
    }
    // ---- per-workgroup totals (the only barrier of the kernel) ----
    const uint32_t wh = wave_inclusive_scan(my_hits), wn = wave_inclusive_scan(my_nmin), wm = wave_max(my_maxlen);
    if ((tid & 63) == 63) {
        s_red[0][tid >> 6] = wh;
        s_red[1][tid >> 6] = wn;
        s_red[2][tid >> 6] = wm;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t h = 0, n = 0, mx = 0;
        for (int i = 0; i < EX_THREADS / 64; ++i) {
            h += s_red[0][i];
            n += s_red[1][i];
            mx = s_red[2][i] > mx ? s_red[2][i] : mx;
        }
        fw.wg_hits[blockIdx.x] = h;
        fw.wg_nmin[blockIdx.x] = n;
        fw.wg_maxlen[blockIdx.x] = mx;
    }
}

// one workgroup: wg_base = exclusive scan of wg_hits; batch totals
__global__ __launch_bounds__(SCAN_THREADS) void hit_scan_kernel(SketchArgs a, FilterWork fw, int recount)
{
    __shared__ uint32_t s_w[SCAN_THREADS / 64 + 1];
    __shared__ uint32_t s_n[SCAN_THREADS / 64], s_m[SCAN_THREADS / 64];
    constexpr int PER = MAX_EX_WG / SCAN_THREADS;
    const int tid = threadIdx.x;
    uint32_t v[PER], run = 0, nmin = 0, mx = 0;
    for (int i = 0; i < PER; ++i) {
        const uint32_t g = (uint32_t)tid * PER + i;
        v[i] = run;
        if (g < fw.ex_grid) {
            run += fw.wg_hits[g];
            nmin += fw.wg_nmin[g];
            const uint32_t m = fw.wg_maxlen[g];
            mx = m > mx ? m : mx;
        }
    }
    uint32_t total;
    const uint32_t before = block_exclusive_scan<SCAN_THREADS / 64>(run, s_w, &total);
    for (int i = 0; i < PER; ++i) {
        const uint32_t g = (uint32_t)tid * PER + i;
        if (g < fw.ex_grid) fw.wg_base[g] = before + v[i];
    }
    const uint32_t wn = wave_inclusive_scan(nmin), wm = wave_max(mx);
    if ((tid & 63) == 63) {
        s_n[tid >> 6] = wn;
        s_m[tid >> 6] = wm;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t n = 0, m = 0;
        for (int i = 0; i < SCAN_THREADS / 64; ++i) {
            n += s_n[i];
            m = s_m[i] > m ? s_m[i] : m;
        }
        *a.n_hits = (unsigned long long)total;
        if (n && !recount) atomicAdd(a.n_minimizers, (unsigned long long)n);
        *fw.max_len = (unsigned long long)m;
    }
}

// one (key,val) per index record of every minimizer, at the scanned offset: hits come out ordered by (read, pos)
__global__ __launch_bounds__(EX_THREADS) void expand_kernel(SketchArgs a, FilterWork fw)
{
    __shared__ uint32_t s_w[EX_THREADS / 64 + 1];
    const int tid = threadIdx.x;
    uint32_t t_begin, t_end;
    candidate_range(fw, blockIdx.x, gridDim.x, t_begin, t_end);
    uint64_t base = fw.wg_base[blockIdx.x];
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += EX_THREADS) {
        const uint32_t t = t0 + tid;
        const uint32_t pos1 = t < t_end ? fw.cand_pos1[t] : 0u;
        uint64_t info = 0;
        uint2 rec = make_uint2(0, 0);
        if (pos1) {
            info = fw.cand_info[t];
            rec = a.slot_rec[(uint32_t)(info >> 32)];
        }
        uint32_t total;
        const uint32_t off = block_exclusive_scan<EX_THREADS / 64>(rec.y, s_w, &total);
        const uint64_t at = base + off;
        if (pos1 && at + rec.y <= a.hit_capacity) {
            const uint32_t read = (uint32_t)info & 0x7FFFFFFFu, strand = ((uint32_t)info >> 31) & 1u;
            for (uint32_t q = 0; q < rec.y; ++q) {
                const uint32_t kn = a.rec_knode[rec.x + q];
                const uint32_t prg = a.rec_prg[rec.x + q];
                const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                a.hit_key[at + q] = pack_hit_key(read, prg, rev, pos1 - 1);
                a.hit_val[at + q] = kn >> 1;
            }
        }
        base += total;
    }
}

// Hits ordered by (read, pos) -> ordered by (read, prg, strand, pos).  A short read's hits nearly always lie in one
// (prg, strand) group, i.e. they are in order already: read_inversion_kernel lists the few reads that are not (one
// thread per adjacent pair, the first inversion of a read reports it) and read_fix_kernel reorders just those, in
// place and stable.  Long reads take the global radix sort instead (Mapper::run_batch).
__global__ void read_inversion_kernel(const uint64_t* __restrict__ key, uint32_t n, uint2* __restrict__ list, uint32_t cap,
    unsigned long long* __restrict__ count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const uint64_t a = key[i], b = key[i + 1];
    const uint32_t read = hit_read(a);
    if (hit_read(b) != read || a <= b) return;
    // rare from here on: [s, e) = the hits of this read; an earlier inversion in it means another thread reports it
    uint32_t s = i;
    while (s > 0 && hit_read(key[s - 1]) == read) {
        if (key[s - 1] > key[s]) return;
        --s;
    }
    uint32_t e = i + 2;
    while (e < n && hit_read(key[e]) == read) ++e;
    const unsigned long long at = atomicAdd(count, 1ull);
    if (at < cap) list[at] = make_uint2(s, e - s); // cap >= n / 2 >= the number of reads with two hits
}

constexpr int RS_THREADS = 256, RS_MAX = 1024;
__global__ __launch_bounds__(RS_THREADS) void read_fix_kernel(uint64_t* __restrict__ key, uint32_t* __restrict__ val,
    const uint2* __restrict__ list, const unsigned long long* __restrict__ count)
{
    __shared__ uint64_t s_key[RS_MAX];
    __shared__ uint32_t s_val[RS_MAX];
    const uint32_t n_list = (uint32_t)*count;
    for (uint32_t r = blockIdx.x; r < n_list; r += gridDim.x) {
        const uint32_t start = list[r].x, len = list[r].y;
        if (len > RS_MAX) { // not expected for short reads; correct but serial
            if (threadIdx.x == 0)
                for (uint32_t j = start + 1; j < start + len; ++j) {
                    const uint64_t kj = key[j];
                    const uint32_t vj = val[j];
                    uint32_t p = j;
                    while (p > start && key[p - 1] > kj) {
                        key[p] = key[p - 1];
                        val[p] = val[p - 1];
                        --p;
                    }
                    key[p] = kj;
                    val[p] = vj;
                }
            continue;
        }
        for (uint32_t t = threadIdx.x; t < len; t += RS_THREADS) {
            s_key[t] = key[start + t];
            s_val[t] = val[start + t];
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < len; t += RS_THREADS) {
            const uint64_t kt = s_key[t];
            uint32_t before = 0;
            for (uint32_t j = 0; j < t; ++j) before += s_key[j] <= kt ? 1u : 0u; // earlier hits precede on ties
            for (uint32_t j = t + 1; j < len; ++j) before += s_key[j] < kt ? 1u : 0u;
            key[start + before] = kt;
            val[start + before] = s_val[t];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// per-read clustering straight from the candidate list
// ---------------------------------------------------------------------------------------------
// The candidates leave verify_count_kernel ordered by (read, position), so all minimizers of a read sit next to each
// other, and a short read has a few dozen hits, nearly always in ONE cluster.  read_cluster_kernel therefore never
// materialises the hit list.  A workgroup stages RC_SLOTS consecutive candidates and their index records (the hits) in
// LDS.  A position gap > max_diff between two consecutive minimizers of a read starts a new segment; as long as all
// hits of the read lie in one (prg, strand) group, the segments ARE the clusters and the overlap sweep of
// cluster_filter_kernel cannot drop any of them (same group, disjoint position ranges).  So every minimizer adds its
// hits to its segment's counter, the first slot of a segment applies the size threshold of cluster_eval_kernel, and
// every minimizer of a kept segment then adds its own hits to the coverage vector -- all of it data parallel.  Only
// the reads whose hits fall into several groups are walked serially by the thread of their first candidate (clusters
// per group split at gaps, size threshold, the overlap sweep; pandora define_clusters / filter_clusters).  This replaces expand + reorder
// + flag + scan + start + eval + filter + count + accumulate (13 launches) for such reads.  A read that does not fit
// (its candidates run past the staged range, more than RC_MAXC clusters, too many hits in the chunk, a position >=
// 2^16) is left alone: its candidates keep cand_pos1 != 0, n_complex counts it, and the host sends what is left
// through the generic pipeline (long reads always go that way).  Handled reads get cand_pos1 = 0.  (A workgroup may
// read cand_pos1 of a neighbouring chunk's read while that chunk zeroes it: either value only moves where the foreign
// hits land in LDS, nothing else.)
constexpr int RC_THREADS = 1024;
constexpr int RC_WAVES = RC_THREADS / 64;
constexpr int RC_PER = 2;
constexpr int RC_SLOTS = RC_THREADS * RC_PER; // staged candidates
constexpr int RC_AHEAD = 512;                 // look-ahead: a read belongs to the chunk that owns its first candidate
constexpr int RC_OWN = RC_SLOTS - RC_AHEAD;
constexpr int RC_HCAP = 3072;                 // staged hits
constexpr int RC_POOL = 64;                   // reads per chunk that may take the wave path
constexpr uint32_t RC_IRREGULAR = 2u, RC_COMPLEX = 1u;

// later (read start << 16 | segment start) pair: the read start decides, then the segment start.  A gap inside a read
// that began in an earlier thread's slots carries read start 0 here: it must take the earlier pair's read start.
__device__ __forceinline__ uint32_t pack_max(uint32_t earlier, uint32_t later)
{
    const uint32_t lead = (later >> 16) > (earlier >> 16) ? (later >> 16) : (earlier >> 16);
    const uint32_t seg = (later & 0xFFFFu) > (earlier & 0xFFFFu) ? (later & 0xFFFFu) : (earlier & 0xFFFFu);
    return (lead << 16) | seg;
}
// LDS traffic only: global loads, stores and atomics stay in flight across the barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(RC_THREADS, 8) void read_cluster_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    extern __shared__ uint32_t s_hist[]; // clusters kept per PRG
    __shared__ uint32_t s_read[RC_SLOTS + 1], s_hstart[RC_SLOTS + 1];
    __shared__ uint16_t s_pos1[RC_SLOTS]; // read position + 1 of a minimizer (0xFFFF: too far for this kernel), 0 = not a minimizer
    __shared__ uint32_t s_gt[RC_SLOTS];   // at a segment's first slot: group << 16 | size threshold; later the decision for the segment
    __shared__ uint16_t s_lead[RC_SLOTS]; // 1 + slot of the first candidate of this slot's read (0: the read started in an earlier chunk)
    __shared__ uint16_t s_seg[RC_SLOTS];  // 1 + first slot of this slot's segment
    __shared__ uint16_t s_end[RC_SLOTS];  // at a segment's first slot: the first slot of the next segment
    __shared__ uint8_t s_cplx[RC_SLOTS], s_irrf[RC_SLOTS]; // at a read's first slot: does not fit / hits in several groups
    __shared__ uint16_t s_grp[RC_HCAP];   // per hit: prg << 1 | rev
    __shared__ uint16_t s_hpos[RC_HCAP];  // per hit: read position
    __shared__ uint32_t s_cov[RC_HCAP];   // per hit: index into the coverage vector, 2 * k-mer node + rev
    __shared__ uint32_t s_w[2][RC_WAVES];
    __shared__ uint32_t s_irr[RC_POOL];
    __shared__ uint32_t s_prev_read, s_n_irr, s_chunk;
    __shared__ unsigned long long s_tot[3];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (*reinterpret_cast<volatile uint32_t*>(a.overflow) & 4u) return; // a candidate slice overflowed: the host re-runs the batch
    const uint32_t total = fw.cand_prefix[fw.n_slices];
    for (uint32_t i = tid; i < rc.n_prgs; i += RC_THREADS) s_hist[i] = 0;
    if (tid < 3) s_tot[tid] = 0;
    unsigned long long my_kept_hits = 0;
    uint32_t my_kept = 0, my_complex = 0;

    // chunks are handed out by a global counter (two or three chunks per workgroup: a static split leaves a third of
    // the workgroups idle for the last round); the next chunk number is fetched while the current one is processed
    if (tid == 0) s_chunk = atomicAdd(rc.chunk_counter, 1u);
    for (;;) {
        lds_barrier(); // LDS of the previous chunk is free, s_chunk is there
        const uint64_t base64 = (uint64_t)s_chunk * RC_OWN;
        if (base64 >= total) break;
        const uint32_t base = (uint32_t)base64;
        const uint32_t n_loaded = total - base < (uint32_t)RC_SLOTS ? total - base : (uint32_t)RC_SLOTS;
        const uint32_t n_own = total - base < (uint32_t)RC_OWN ? total - base : (uint32_t)RC_OWN;
        lds_barrier(); // everybody has read s_chunk
        if (tid == 0) s_chunk = atomicAdd(rc.chunk_counter, 1u);
        // ---- A: stage the candidates: three coalesced loads per slot, nothing depends on them but LDS work ----
        uint4 crec[RC_PER];
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            uint32_t read = READ_NONE, pos1 = 0;
            crec[q] = make_uint4(0, 0, 0, 0);
            if (i < n_loaded) {
                read = (uint32_t)fw.cand_info[base + i] & 0x7FFFFFFFu;
                pos1 = fw.cand_pos1[base + i];
                crec[q] = fw.cand_rec[base + i];
                if (!pos1) crec[q].y = 0; // handled by another chunk in the meantime (look-ahead slots only)
            }
            s_read[i] = read;
            s_pos1[i] = (uint16_t)(pos1 < 0xFFFFu ? pos1 : 0xFFFFu);
            s_hstart[i] = crec[q].y;
            s_cplx[i] = 0;
            s_irrf[i] = 0;
        }
        if (tid == 0) {
            s_prev_read = base ? ((uint32_t)fw.cand_info[base - 1] & 0x7FFFFFFFu) : 0xFFFFFFFFu;
            // the candidate after the staged range: a read that runs on into it does not fit
            s_read[RC_SLOTS] = (n_loaded == (uint32_t)RC_SLOTS && base + n_loaded < total) ? ((uint32_t)fw.cand_info[base + n_loaded] & 0x7FFFFFFFu)
                                                                                           : 0xFFFFFFFFu;
            s_n_irr = 0;
        }
        lds_barrier();
        // ---- B: exclusive sum scan of the hit counts and ONE inclusive max scan of (read start << 16 | segment start), in slot
        // order: both starts only grow along the slots and a read start is a segment start, so the packed maximum is the pair ----
        {
            uint32_t v[RC_PER], m[RC_PER], run = 0, mx = 0;
#pragma unroll
            for (int q = 0; q < RC_PER; ++q) {
                const uint32_t i = (uint32_t)tid * RC_PER + q;
                v[q] = run;
                run += s_hstart[i];
                const uint32_t r = s_read[i], p1 = s_pos1[i];
                const uint32_t r_prev = i ? s_read[i - 1] : s_prev_read;
                if (r != r_prev) {
                    if (r != READ_NONE) mx = ((i + 1) << 16) | (i + 1);
                } else if (p1 && i) { // a gap to the previous minimizer of the read?
                    uint32_t pp = s_pos1[i - 1];
                    if (!pp) { // rare: candidates that are no minimizers lie between
                        int j = (int)i - 2;
                        while (j >= 0 && s_read[j] == r && !s_pos1[j]) --j;
                        pp = (j >= 0 && s_read[j] == r) ? s_pos1[j] : p1;
                    }
                    if ((int)(p1 - pp) > rc.max_diff) mx = (mx & 0xFFFF0000u) | (i + 1);
                }
                m[q] = mx;
            }
            uint32_t incl = run, imx = mx;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t n = __shfl_up(incl, off), x = __shfl_up(imx, off);
                if (lane >= off) {
                    incl += n;
                    imx = pack_max(x, imx);
                }
            }
            if (lane == 63) {
                s_w[0][wave] = incl;
                s_w[1][wave] = imx;
            }
            const uint32_t excl_mx_in_wave = __shfl_up(imx, 1);
            lds_barrier();
            uint32_t before = incl - run, mx_before = lane ? excl_mx_in_wave : 0u, sum = 0;
#pragma unroll
            for (int i = 0; i < RC_WAVES; ++i) {
                const uint32_t x = s_w[0][i], y = s_w[1][i];
                if (i < wave) {
                    before += x;
                    mx_before = pack_max(mx_before, y);
                }
                sum += x;
            }
#pragma unroll
            for (int q = 0; q < RC_PER; ++q) {
                const uint32_t i = (uint32_t)tid * RC_PER + q;
                const uint32_t pm = pack_max(mx_before, m[q]);
                s_hstart[i] = before + v[q];
                s_lead[i] = (uint16_t)(pm >> 16);
                s_seg[i] = (uint16_t)pm;
            }
            if (tid == 0) s_hstart[RC_SLOTS] = sum;
        }
        lds_barrier();
        // ---- C: the hits (one per index record of every minimizer); every segment start closes the segment before it ----
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const uint32_t seg = s_seg[i];
            if (seg == i + 1 && i > 0 && s_seg[i - 1]) s_end[s_seg[i - 1] - 1] = (uint16_t)i;
            if (i == RC_SLOTS - 1 && seg) s_end[seg - 1] = (uint16_t)RC_SLOTS;
            const uint32_t cnt = crec[q].y;
            if (!cnt) continue;
            const uint32_t h0 = s_hstart[i], lead = s_lead[i];
            const uint32_t pos = (uint32_t)s_pos1[i] - 1, strand = crec[q].z >> 31;
            if (h0 + cnt > (uint32_t)RC_HCAP || pos >= 0xFFFEu) {
                if (lead && lead <= n_own) s_cplx[lead - 1] = 1;
                continue;
            }
            s_grp[h0] = (uint16_t)((crec[q].z >> 16) & 0x7FFFu);
            s_hpos[h0] = (uint16_t)pos;
            s_cov[h0] = crec[q].w;
            for (uint32_t r = 1; r < cnt; ++r) { // rare: a k-mer that several k-mer nodes share
                const uint32_t kn = a.rec_knode[crec[q].x + r], prg = a.rec_prg[crec[q].x + r];
                const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                s_grp[h0 + r] = (uint16_t)((prg << 1) | rev);
                s_hpos[h0 + r] = (uint16_t)pos;
                s_cov[h0 + r] = (kn >> 1) * 2u + rev;
            }
        }
        lds_barrier();
        // ---- D: a minimizer whose group differs from the previous one of its read makes the read irregular; the first
        // minimizer of a segment names the segment's group and threshold ----
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const uint32_t lead = s_lead[i], cnt = crec[q].y;
            if (!cnt || lead == 0 || lead > n_own) continue;
            const uint32_t first = lead - 1, seg = (uint32_t)s_seg[i] - 1, h0 = s_hstart[i];
            if (h0 + cnt > (uint32_t)RC_HCAP) continue; // complex already
            const uint32_t g = (crec[q].z >> 16) & 0x7FFFu;
            bool irregular = false;
            for (uint32_t r = 1; r < cnt; ++r) irregular |= s_grp[h0 + r] != g;
            int j = (int)i - 1; // the previous minimizer of the read
            while (j >= (int)first && !s_pos1[j]) --j;
            if (j >= (int)first && s_grp[s_hstart[j]] != g) irregular = true; // (s_hstart[j] <= h0 < RC_HCAP)
            if (j < (int)seg) s_gt[seg] = crec[q].z & 0x7FFFFFFFu;
            if (irregular) s_irrf[first] = 1;
        }
        if (tid == 0 && s_read[RC_SLOTS] == s_read[RC_SLOTS - 1]) { // the last staged read runs on past the staged range
            const uint32_t lead = s_lead[RC_SLOTS - 1];
            if (lead && lead <= n_own) s_cplx[lead - 1] = 1;
        }
        lds_barrier();
        // ---- E: the first slot of every segment decides for the segment; reads with several groups queue for the wave path ----
        uint32_t dec[RC_PER];
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const uint32_t lead = s_lead[i];
            dec[q] = 0xFFFFFFFFu; // not the first slot of a segment
            if (i >= n_loaded || s_seg[i] != i + 1 || lead == 0 || lead > n_own) continue;
            uint32_t flags = (s_cplx[lead - 1] ? RC_COMPLEX : 0u) | (s_irrf[lead - 1] ? RC_IRREGULAR : 0u);
            if (lead == i + 1) { // first slot of the read
                if (flags & RC_COMPLEX) ++my_complex;
                else if (flags & RC_IRREGULAR) {
                    uint32_t e = i; // the first slot after the read: follow its segments
                    for (uint32_t sg = i; sg < (uint32_t)RC_SLOTS && s_lead[sg] == lead; sg = s_end[sg]) e = s_end[sg];
                    const uint32_t at = atomicAdd(&s_n_irr, 1u);
                    if (at < (uint32_t)RC_POOL) s_irr[at] = i | (e << 16);
                    else ++my_complex;
                }
            }
            const uint32_t n_hits = s_hstart[s_end[i]] - s_hstart[i];
            dec[q] = 0; // 0 leave alone, 1 handled, 2 handled and every hit counts
            if (!flags && n_hits) {
                const uint32_t g_thr = s_gt[i];
                dec[q] = 1;
                if (n_hits > (g_thr & 0xFFFFu)) {
                    dec[q] = 2;
                    atomicAdd(&s_hist[g_thr >> 17], 1u);
                    ++my_kept;
                    my_kept_hits += n_hits;
                }
            }
        }
        lds_barrier(); // all thresholds are read
#pragma unroll
        for (int q = 0; q < RC_PER; ++q)
            if (dec[q] != 0xFFFFFFFFu) s_gt[(uint32_t)tid + (uint32_t)q * RC_THREADS] = dec[q];
        lds_barrier();
        // ---- F: the minimizers of the kept segments ----
        {
#pragma unroll
            for (int q = 0; q < RC_PER; ++q) {
                const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
                const uint32_t lead = s_lead[i], cnt = crec[q].y;
                if (!cnt || lead == 0 || lead > n_own) continue;
                const uint32_t decision = s_gt[(uint32_t)s_seg[i] - 1];
                if (!decision) continue;
                fw.cand_pos1[base + i] = 0; // handled
                if (decision == 2) {
                    atomicAdd(&rc.covg[crec[q].w], 1u);
                    const uint32_t h0 = s_hstart[i];
                    for (uint32_t r = 1; r < cnt; ++r) atomicAdd(&rc.covg[s_cov[h0 + r]], 1u);
                }
            }
        }
        // ---- G: reads with hits in several groups, one wave per read: lane j holds cluster j, the hits are broadcast one by one
        // (clusters per group split at gaps, size threshold, the overlap sweep of cluster_filter_kernel) ----
        const uint32_t n_irr = s_n_irr < (uint32_t)RC_POOL ? s_n_irr : (uint32_t)RC_POOL;
        for (uint32_t r = wave; r < n_irr; r += RC_WAVES) {
            const uint32_t i = s_irr[r] & 0xFFFFu, e = s_irr[r] >> 16;
            const uint32_t read = s_read[i], hb = s_hstart[i], he = s_hstart[e];
            uint32_t cl_g = 0, cl_n = 0, cl_first = 0, cl_last = 0;
            int nc = 0;
            bool complex = false;
            for (uint32_t b = hb; b < he && !complex; b += 64) {
                const uint32_t h = b + lane;
                const uint32_t hg = h < he ? s_grp[h] : 0u, hp = h < he ? s_hpos[h] : 0u;
                const int nb = he - b < 64u ? (int)(he - b) : 64;
                for (int t = 0; t < nb; ++t) {
                    const uint32_t g = __builtin_amdgcn_readlane(hg, t), pos = __builtin_amdgcn_readlane(hp, t);
                    const uint64_t mm = __ballot(lane < nc && cl_g == g);
                    if (mm) {
                        const int f = 63 - __clzll((long long)mm);
                        const uint32_t last_f = __shfl(cl_last, f);
                        if ((int)(pos - last_f) <= rc.max_diff) {
                            if (lane == f) {
                                ++cl_n;
                                cl_last = pos;
                            }
                            continue;
                        }
                    }
                    if (nc == 64) { complex = true; break; }
                    if (lane == nc) {
                        cl_g = g;
                        cl_n = 1;
                        cl_first = cl_last = pos;
                    }
                    ++nc;
                }
            }
            if (complex) {
                if (lane == 0) ++my_complex;
                continue;
            }
            const uint64_t len = a.offsets[read + 1] - a.offsets[read];
            const uint64_t expected = len * 2 / (uint64_t)(a.w + 1);
            bool kept = false;
            if (lane < nc) {
                uint64_t m = rc.prg_min_path_len[cl_g >> 1];
                if (expected < m) m = expected;
                const uint32_t length_based = (uint32_t)((double)m * rc.fraction);
                const uint32_t thr = length_based > rc.min_cluster_size ? length_based : rc.min_cluster_size;
                kept = cl_n > thr;
            }
            uint64_t alive = __ballot(kept);
            if (alive & (alive - 1)) {
                // rank in cluster order: first position, larger first, prg, forward first
                uint32_t rank = 0;
                for (uint64_t mm = alive; mm; mm &= mm - 1) {
                    const int o = __ffsll((long long)mm) - 1;
                    const uint32_t of = __shfl(cl_first, o), on = __shfl(cl_n, o), og = __shfl(cl_g, o);
                    rank += (of < cl_first || (of == cl_first && (on > cl_n || (on == cl_n && og < cl_g)))) ? 1u : 0u;
                }
                const int nk = __popcll(alive);
                int prev = -1;
                uint32_t pg = 0, pn = 0, p_last = 0;
                for (int o = 0; o < nk; ++o) {
                    const int cur = __ffsll((long long)__ballot(kept && rank == (uint32_t)o)) - 1;
                    const uint32_t cg = __shfl(cl_g, cur), cn = __shfl(cl_n, cur), c_last = __shfl(cl_last, cur);
                    if (prev >= 0) {
                        const bool same_prg_other_strand = (pg >> 1) == (cg >> 1) && (pg & 1u) != (cg & 1u);
                        if (same_prg_other_strand || c_last <= p_last) {
                            if (pn >= cn) {
                                alive &= ~(1ull << cur);
                                continue;
                            }
                            alive &= ~(1ull << prev);
                        }
                    }
                    prev = cur;
                    pg = cg;
                    pn = cn;
                    p_last = c_last;
                }
            }
            if ((alive >> lane) & 1ull) {
                atomicAdd(&s_hist[cl_g >> 1], 1u);
                ++my_kept;
                my_kept_hits += cl_n;
            }
            for (uint32_t b = hb; b < he; b += 64) { // every hit finds its cluster among the survivors
                const uint32_t h = b + lane;
                const uint32_t hg = h < he ? s_grp[h] : 0xFFFFFFFFu, hp = h < he ? s_hpos[h] : 0u;
                for (uint64_t mm = alive; mm; mm &= mm - 1) {
                    const int o = __ffsll((long long)mm) - 1;
                    const uint32_t og = __shfl(cl_g, o), of = __shfl(cl_first, o), ol = __shfl(cl_last, o);
                    if (hg == og && hp >= of && hp <= ol) atomicAdd(&rc.covg[s_cov[h]], 1u);
                }
            }
            for (uint32_t c = i + lane; c < e; c += 64)
                if (s_pos1[c]) fw.cand_pos1[base + c] = 0; // handled
        }
    }
    // ---- workgroup totals ----
    if (my_kept) atomicAdd(&s_tot[0], (unsigned long long)my_kept);
    if (my_kept_hits) atomicAdd(&s_tot[1], my_kept_hits);
    if (my_complex) atomicAdd(&s_tot[2], (unsigned long long)my_complex);
    __syncthreads();
    for (uint32_t i = tid; i < rc.n_prgs; i += RC_THREADS)
        if (s_hist[i]) atomicAdd(&rc.prg_reads[i], s_hist[i]);
    if (tid == 0) {
        if (s_tot[0]) atomicAdd(rc.n_clusters_kept, s_tot[0]);
        if (s_tot[1]) atomicAdd(rc.n_hits_kept, s_tot[1]);
        if (s_tot[2]) atomicAdd(rc.n_complex, s_tot[2]);
    }
}

// what read_cluster_kernel left behind: hits and longest read per workgroup range (the layout verify_count_kernel wrote)
__global__ __launch_bounds__(EX_THREADS) void recount_kernel(SketchArgs a, FilterWork fw)
{
    __shared__ uint32_t s_red[2][EX_THREADS / 64];
    const int tid = threadIdx.x;
    uint32_t t_begin, t_end;
    candidate_range(fw, blockIdx.x, gridDim.x, t_begin, t_end);
    uint32_t my_hits = 0, my_maxlen = 0;
    for (uint32_t t = t_begin + tid; t < t_end; t += EX_THREADS) {
        if (!fw.cand_pos1[t]) continue;
        const uint64_t info = fw.cand_info[t];
        my_hits += a.slot_rec[(uint32_t)(info >> 32)].y;
        const uint32_t read = (uint32_t)info & 0x7FFFFFFFu;
        const uint64_t len64 = a.offsets[read + 1] - a.offsets[read];
        const uint32_t len = len64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len64;
        my_maxlen = len > my_maxlen ? len : my_maxlen;
    }
    const uint32_t wh = wave_inclusive_scan(my_hits), wm = wave_max(my_maxlen);
    if ((tid & 63) == 63) {
        s_red[0][tid >> 6] = wh;
        s_red[1][tid >> 6] = wm;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t h = 0, mx = 0;
        for (int i = 0; i < EX_THREADS / 64; ++i) {
            h += s_red[0][i];
            mx = s_red[1][i] > mx ? s_red[1][i] : mx;
        }
        fw.wg_hits[blockIdx.x] = h;
        fw.wg_nmin[blockIdx.x] = 0;
        fw.wg_maxlen[blockIdx.x] = mx;
    }
}

// DRPRG_FT_DEBUG=8: no read_cluster_kernel; the generic pipeline runs iff there is a hit
__global__ void flag_complex_kernel(const unsigned long long* n_hits, unsigned long long* n_complex)
{
    if (*n_hits) *n_complex = 1;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
uint32_t filter_n_tiles(uint64_t n_bases) { return (uint32_t)((n_bases + FT_WPOS - 1) / FT_WPOS); }

uint32_t filter_grid(bool level0, int n_cus, uint32_t n_tiles)
{
    // persistent grid: the workgroups that stay resident (1024 threads each; 64 KB of LDS: two per CU, 128 KB with
    // level 0: one per CU), never more waves than there are wave tiles
    uint32_t grid = (uint32_t)n_cus * (level0 ? 1 : 2);
    const uint32_t need = (n_tiles + FT_WAVES - 1) / FT_WAVES;
    if (grid > need) grid = need;
    if (grid > (uint32_t)(MAX_SLICES / (FT_WAVES * FT_SUB))) grid = MAX_SLICES / (FT_WAVES * FT_SUB);
    return grid ? grid : 1;
}

size_t filter_small_words() { return (size_t)MAX_SLICES * 3 + 1 + 4 * (size_t)MAX_EX_WG; }

hipError_t launch_sketch_filter(const SketchArgs& a, const BloomTables& bt, int n_cus, const FilterBuffers& b, const ReadClusterArgs& rc,
    FilterWork& fw, hipStream_t stream, KernelTimer timer)
{
    fw = FilterWork {};
    if (a.n_bases == 0) return hipSuccess;
    if ((1u << bt.bloom_wbits) > (uint32_t)FT_BLOOM_WORDS) return hipErrorInvalidValue;
    if (bt.bloom0 && (1u << bt.bloom0_wbits) != (uint32_t)FT_L0_WORDS) return hipErrorInvalidValue;
    if (bt.bloom0 && (!b.raw_grp || !bt.bloomr)) return hipErrorInvalidValue;
    if (const char* dbg = std::getenv("DRPRG_FT_DEBUG")) fw.debug = (uint32_t)std::atoi(dbg); // 1 no filter test, 4 no level 0, 8 no read_cluster_kernel
    const bool level0 = bt.bloom0 != nullptr && a.k == 15 && ((size_t)4 << bt.bloom_wbits) + (size_t)FT_L0_WORDS * 4 <= 160 * 1024
        && !(fw.debug & 4u);
    fw.bloom = bt.bloom;
    fw.bloom_wbits = bt.bloom_wbits;
    fw.bloomr = bt.bloomr;
    fw.bloom0 = level0 ? bt.bloom0 : nullptr;
    fw.bloom0_wbits = level0 ? bt.bloom0_wbits : 0;
    fw.n_tiles = filter_n_tiles(a.n_bases);
    const uint32_t grid = filter_grid(level0, n_cus, fw.n_tiles);
    fw.n_slices = grid * FT_WAVES * FT_SUB;
    fw.tiles_per_wave = (fw.n_tiles + grid * FT_WAVES - 1) / (grid * FT_WAVES);
    fw.tiles_per_slice = (fw.tiles_per_wave + FT_SUB - 1) / FT_SUB;
    fw.raw_slice = (uint32_t)std::min<uint64_t>(b.raw_capacity / fw.n_slices, 0x7FFFFFFFull / fw.n_slices);
    fw.raw_pos = b.raw_pos;
    fw.cand_info = b.cand_info;
    fw.cand_pos1 = b.cand_pos1;
    fw.cand_rec = b.cand_rec;
    fw.slice_count = b.small;
    fw.cand_prefix = b.small + MAX_SLICES;
    fw.grp_count = b.small + 2 * MAX_SLICES + 1;
    fw.raw_grp = b.raw_grp;
    fw.wg_hits = b.small + 3 * MAX_SLICES + 1;
    fw.wg_nmin = fw.wg_hits + MAX_EX_WG;
    fw.wg_maxlen = fw.wg_nmin + MAX_EX_WG;
    fw.wg_base = fw.wg_maxlen + MAX_EX_WG;
    fw.ex_grid = std::min<uint32_t>((uint32_t)n_cus * 8, MAX_EX_WG);
    fw.max_len = b.max_len;
    if (timer.begin) HIP_TRY(hipEventRecord(timer.begin, stream));
    {
        using Kernel = void (*)(SketchArgs, FilterWork);
        const int which = level0 ? 2 : (a.k < 12 ? 1 : 0);
        const Kernel kernel = which == 2 ? &sketch_filter_kernel<false, true>
            : which == 1                 ? &sketch_filter_kernel<true, false>
                                         : &sketch_filter_kernel<false, false>;
        const size_t dyn = level0 ? (size_t)FT_L0_WORDS * 4 : ((size_t)4 << bt.bloom_wbits);
        static size_t configured[3] = { 0, 0, 0 };
        if (dyn > configured[which]) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            configured[which] = dyn;
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(FT_THREADS), dyn, stream, a, fw);
    }
    HIP_TRY(hipGetLastError());
    if (timer.end) HIP_TRY(hipEventRecord(timer.end, stream));
    if (level0) {
        static size_t refine_configured = 0;
        const size_t dyn = (size_t)4 << BLOOMR_WBITS;
        if (dyn > refine_configured) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            refine_configured = dyn;
        }
        hipLaunchKernelGGL(refine_kernel, dim3(std::min<uint32_t>((fw.n_slices + RF_THREADS / 64 - 1) / (RF_THREADS / 64), (uint32_t)n_cus * 2)), dim3(RF_THREADS), dyn, stream, a, fw);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(cand_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, fw);
    hipLaunchKernelGGL(cand_gather_kernel, dim3(fw.n_slices), dim3(64), 0, stream, fw);
    hipLaunchKernelGGL(verify_count_kernel, dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw, rc);
    hipLaunchKernelGGL(hit_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, a, fw, 0);
    HIP_TRY(hipGetLastError());
    if (fw.debug & 8u) { // every read with a hit goes the generic way
        hipLaunchKernelGGL(HIP_KERNEL_NAME(flag_complex_kernel), dim3(1), dim3(1), 0, stream, a.n_hits, rc.n_complex);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(read_cluster_kernel, dim3((uint32_t)n_cus * 2), dim3(RC_THREADS), (size_t)rc.n_prgs * sizeof(uint32_t), stream, a, fw, rc);
    return hipGetLastError();
}

hipError_t launch_filter_recount(const SketchArgs& a, const FilterWork& fw, hipStream_t stream)
{
    hipLaunchKernelGGL(recount_kernel, dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(hit_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, a, fw, 1);
    return hipGetLastError();
}

hipError_t launch_filter_expand(const SketchArgs& a, const FilterWork& fw, hipStream_t stream)
{
    hipLaunchKernelGGL(expand_kernel, dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw);
    return hipGetLastError();
}

hipError_t launch_read_sort(uint64_t* key, uint32_t* val, uint32_t n, uint32_t* scratch, uint64_t scratch_words, unsigned long long* count,
    hipStream_t stream)
{
    if (n < 2) return hipSuccess;
    uint2* list = reinterpret_cast<uint2*>(scratch);
    hipLaunchKernelGGL(read_inversion_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, key, n, list, (uint32_t)(scratch_words / 2), count);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(read_fix_kernel, dim3(128), dim3(RS_THREADS), 0, stream, key, val, list, count);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
