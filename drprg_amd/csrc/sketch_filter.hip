// sketch_filter.hip -- K1+K2 in their fast form: persistent waves + an LDS-resident Bloom prefilter.
//
// A read minimizer can only produce a hit if its k-mer is an index k-mer (in either orientation).  So instead of
// hashing every k-mer of every read (sketch_probe.hip: ~86 VALU instructions per base, two 15-op hashes each):
//
//   sketch_filter_kernel   streams the concatenated base buffer through persistent waves, packs it to 2 bits per base
//                          in registers and tests each position's k-mer *code* against a Bloom filter of the index
//                          k-mer codes that stays in LDS for the lifetime of the workgroup; the positions that pass
//                          (index k-mers + ~0.02 % false positives) are appended, without any global atomic, to the
//                          workgroup's slice of a candidate buffer.  The loop has no barrier and waits on nothing but
//                          its own loads: three tiles of bases are in flight per wave in a register ring, the right
//                          neighbour's packed word arrives through a DPP wave shift.
//   verify_expand_kernel   one lane per candidate, start to finish: canonical hash from the raw bases -> exact table
//                          lookup (false positives end here) -> read lookup -> window-minimizer test over the 2w-1
//                          neighbouring k-mers inside the read, hashed one by one out of a register shift register
//                          -> one (key,val) hit per index record, output space reserved once per 256 candidates.
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
constexpr int FT_G = 16;                // positions per lane
constexpr int FT_WPOS = 63 * FT_G;      // positions per wave tile: lane 63's word is only lane 62's right neighbour
constexpr int FT_BLOOM_WORDS = 1 << 14; // static LDS: the filter sits at LDS address 0, so a hash is an address

// 16 ASCII bases -> 32 bits, 2 per base, first base in the lowest bits.  The 2-bit letter is bits 2:1 of the ASCII code
// (A 0, C 1, T 2, G 3, either case; anything else aliases one of them: it can only create a false candidate, which
// verify_expand_kernel rejects from the raw bases).  One multiply gathers the four fields of a dword into its top byte.
__device__ inline uint32_t pack16le(const uint4& in)
{
    constexpr uint32_t M = (1u << 23) | (1u << 17) | (1u << 11) | (1u << 5);
    const uint32_t p0 = (in.x & 0x06060606u) * M, p1 = (in.y & 0x06060606u) * M;
    const uint32_t p2 = (in.z & 0x06060606u) * M, p3 = (in.w & 0x06060606u) * M;
    // byte 3 of p0..p3 -> bytes 0..3
    return __builtin_amdgcn_perm(p1, p0, 0x0c0c0703u) | __builtin_amdgcn_perm(p3, p2, 0x07030c0cu);
}

struct FilterArgs {
    const uint32_t* bloom;
    uint32_t bloom_wbits;
    uint32_t n_tiles;    // wave tiles of FT_WPOS positions
    uint64_t* raw_pos;   // [grid][raw_slice]: global base position of a candidate k-mer
    uint32_t* raw_count; // [grid]
    uint32_t raw_slice;
    uint32_t debug; // ablation switch for profiling (DRPRG_FT_DEBUG): 1 = skip the Bloom test
};

// Bloom filter layout (built by FlatIndex, index.cpp; both orientations of every index k-mer are entered):
//   code   = little-endian 2-bit letters of the k-mer (pack16le alphabet), 2k bits
//   level 1, keyed on the first min(k,12) bases: x = code & 0xFFFFFF, h = (x * BLOOM_C1) mod 2^32,
//            word (h >> 18) & (words-1), bits 31-(h & 31), 31-((h >> 8) & 31), 31-((x >> 16) & 31)
//   level 2, keyed on the whole code: h2 = code * BLOOM_C2, word h2 >> (32-wbits), bits h2 & 31, (h2>>5) & 31, (h2>>10) & 31
// Level 1 costs 7 VALU + 1 LDS instruction per position (24-bit multiply, byte-select shifts, 3-input AND, funnel-shift
// accumulate); level 2 runs only for the ~1.5 % level-1 survivors.
template <bool SHORT_K> // SHORT_K: k < 12, the level-1 key must be masked to 2k bits
__global__ __launch_bounds__(FT_THREADS) void sketch_filter_kernel(SketchArgs a, FilterArgs fa)
{
    __shared__ uint32_t s_bloom[FT_BLOOM_WORDS];
    extern __shared__ uint32_t s_nraw[]; // one cursor, behind the filter

    const int tid = threadIdx.x, lane = tid & 63;
    const int k = a.k;
    const int64_t n_bases = (int64_t)a.n_bases;
    const uint32_t n_tiles = fa.n_tiles;
    const uint32_t n_words = 1u << fa.bloom_wbits;
    const uint32_t amask = (n_words - 1) << 2; // byte address of the level-1 word = (h >> 16) & amask
    const uint32_t kmask = (k < 16) ? ((1u << (2 * k)) - 1) : 0xFFFFFFFFu;
    const uint32_t kmask24 = kmask & 0xFFFFFFu;
    const int sh_w = 32 - (int)fa.bloom_wbits;

    for (uint32_t i = tid; i < n_words; i += FT_THREADS) s_bloom[i] = fa.bloom[i];
    if (tid == 0) s_nraw[0] = 0;

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
    auto fetch = [&](uint32_t t, uint4& p) {
        if (t < n_tiles) p = load16((int64_t)t * FT_WPOS + (int64_t)lane * 16);
    };
    // register ring: the tile being processed plus three in flight (Little's law: ~48 KB per CU must be outstanding
    // to keep HBM busy; a wave tile is 1 KB, 32 waves per CU)
    const uint32_t stride = gridDim.x * FT_WAVES;
    uint32_t tile = blockIdx.x * FT_WAVES + (uint32_t)(tid >> 6);
    uint4 cur {}, p1 {}, p2 {}, p3 {};
    fetch(tile, cur);
    fetch(tile + stride, p1);
    fetch(tile + 2 * stride, p2);
    __syncthreads(); // Bloom filter in place; the only barrier before the end

    uint64_t* out = fa.raw_pos + (size_t)blockIdx.x * fa.raw_slice;
    for (; tile < n_tiles; tile += stride) {
        fetch(tile + 3 * stride, p3); // stays in flight for three iterations
        const uint32_t w0 = pack16le(cur);
        const uint32_t w1 = __builtin_amdgcn_update_dpp(0u, w0, 0x130 /* wave_shl:1: lane i <- lane i+1 */, 0xF, 0xF, false);
        uint32_t cand = 0;
        if (!(fa.debug & 1u)) {
            // ---- level 1 over my 16 positions; position j ends up in bit j of cand ----
#pragma unroll
            for (int j = FT_G - 1; j >= 0; --j) {
                uint32_t x = j ? __builtin_amdgcn_alignbit(w1, w0, 2 * j) : w0; // code in the low bits, later bases above
                if (SHORT_K) x &= kmask24;
                const uint32_t h = __umul24(x, BLOOM_C1);
                const uint32_t word = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_bloom) + ((h >> 16) & amask));
                uint32_t t2; // word << h[12:8]: the byte select is free, the compiler does not find it
                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                    : "=v"(t2)
                    : "v"(h), "v"(word));
                const uint32_t t = (word << (h & 31)) & t2 & (word << ((x >> 16) & 0xFF)); // bit 31 = all three bits set
                cand = __builtin_amdgcn_alignbit(cand, t, 31);                             // cand = cand << 1 | t >> 31
            }
            if (lane == 63) cand = 0;
            // ---- level 2, only for the survivors: three more bits in a second word, keyed on the whole code ----
            uint32_t c1 = cand;
            cand = 0;
            while (c1) {
                const int j = __ffs(c1) - 1;
                c1 &= c1 - 1;
                const uint32_t f = __funnelshift_r(w0, w1, 2 * j) & kmask;
                const uint32_t h2 = f * BLOOM_C2;
                const uint32_t word = s_bloom[h2 >> sh_w];
                cand |= ((word >> (h2 & 31)) & (word >> ((h2 >> 5) & 31)) & (word >> ((h2 >> 10) & 31)) & 1u) << j;
            }
        }
        // ---- append candidate positions to this workgroup's slice (plain stores, LDS cursor) ----
        if (cand) {
            const int np = __popc(cand);
            uint32_t at = atomicAdd(&s_nraw[0], (uint32_t)np);
            const uint64_t base = (uint64_t)tile * FT_WPOS + (uint64_t)lane * FT_G;
            while (cand) {
                const int j = __ffs(cand) - 1;
                cand &= cand - 1;
                if (at < fa.raw_slice) out[at] = base + (uint64_t)j;
                ++at;
            }
        }
        cur = p1;
        p1 = p2;
        p2 = p3;
    }
    __syncthreads();
    if (tid == 0) {
        fa.raw_count[blockIdx.x] = s_nraw[0];
        if (s_nraw[0] > fa.raw_slice) atomicOr(a.overflow, 4u);
    }
}

// 16 ASCII bases -> packed codes + 16-bit "not ACGT" mask (bit i = base i)
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

constexpr int EX_MAX_WG = 1024;
constexpr int EX_THREADS = 1024;

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

// One lane per candidate, start to finish: the 64 bases around it are packed into registers once (2 bits per base,
// first base highest), the candidate's canonical hash goes to the exact table lookup (Bloom false positives end there),
// then the 2w-1 neighbouring k-mers are hashed one after the other out of a 96-bit shift register -- a third of the
// instructions of giving every neighbour its own lane, each of which had to load and pack its own bases.  It is a read
// minimizer iff the run of neighbours with hash >= its own (inside the read, no N) reaches w-1 across both sides.
// Output space is reserved once per 256 candidates.
__global__ __launch_bounds__(EX_THREADS) void verify_expand_kernel(SketchArgs a, FilterArgs fa, uint32_t n_wg)
{
    using Tr = HashTraits<uint32_t>;
    __shared__ uint32_t s_prefix[EX_MAX_WG + 1];
    __shared__ uint32_t s_part[EX_THREADS];
    __shared__ uint32_t s_cnt, s_nmin;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    // exclusive prefix sums of the (clamped) slice counts: EX_PER entries per thread, then a block scan
    {
        constexpr int EX_PER = EX_MAX_WG / EX_THREADS;
        uint32_t v[EX_PER], run = 0;
        for (int i = 0; i < EX_PER; ++i) {
            const uint32_t b = (uint32_t)tid * EX_PER + i;
            const uint32_t n = b < n_wg ? fa.raw_count[b] : 0u;
            v[i] = run;
            run += n < fa.raw_slice ? n : fa.raw_slice;
        }
        s_part[tid] = run;
        __syncthreads();
        for (int off = 1; off < EX_THREADS; off <<= 1) {
            const uint32_t add = tid >= off ? s_part[tid - off] : 0u;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const uint32_t before = tid ? s_part[tid - 1] : 0u;
        for (int i = 0; i < EX_PER; ++i) s_prefix[tid * EX_PER + i] = before + v[i];
        if (tid == EX_THREADS - 1) s_prefix[EX_MAX_WG] = s_part[EX_THREADS - 1];
        if (tid == 0) { s_cnt = 0; s_nmin = 0; }
        __syncthreads();
    }
    const uint32_t total = s_prefix[EX_MAX_WG];
    const uint32_t* __restrict__ slot_key = reinterpret_cast<const uint32_t*>(a.slot_key);
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const int k = a.k, w = a.w;
    const int sh_k = 32 - 2 * k;
    const uint32_t kmask = (1u << (2 * k)) - 1;
    const int64_t n_bases = (int64_t)a.n_bases;
    const uint32_t per_wg = (total + gridDim.x - 1) / gridDim.x;
    const uint32_t t_begin = blockIdx.x * per_wg;
    const uint32_t t_end = t_begin + per_wg < total ? t_begin + per_wg : total;
    const double reads_per_base = (double)a.n_reads / (double)(a.n_bases ? a.n_bases : 1);
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += EX_THREADS) {
        const uint32_t t = t0 + tid;
        uint32_t cnt = 0, my_off = 0, slot = 0, read = 0, strand = 0;
        uint64_t pos = 0;
        if (t < t_end) {
            uint32_t lo = 0, hi = EX_MAX_WG; // s_prefix[lo] <= t < s_prefix[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= t) lo = mid; else hi = mid;
            }
            const int64_t gp = (int64_t)fa.raw_pos[(size_t)lo * fa.raw_slice + (t - s_prefix[lo])];
            if (gp + k <= n_bases) {
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
                    uint32_t s = table_slot_dev((uint64_t)h, a.table_bits);
                    while (true) {
                        const uint32_t key = slot_key[s];
                        if (key == h) { found = true; break; }
                        if (key == Tr::EMPTY) break;
                        s = (s + 1) & tmask;
                    }
                    slot = s;
                }
                if (found) {
                    // interpolated first guess: exact for fixed-length reads, a short gallop otherwise
                    const uint32_t guess = (uint32_t)((double)gp * reads_per_base);
                    read = find_read_near(a.offsets, a.n_reads, guess, (uint64_t)gp);
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
                        bad >>= of;
                        uint32_t streak = 0, right = 0, alive = 1;
                        for (int i = 0; i < 2 * w - 1; ++i) {
                            const uint32_t f = r0w >> sh_k;
                            r0w = __funnelshift_l(r1w, r0w, 2);
                            r1w = __funnelshift_l(r2w, r1w, 2);
                            r2w <<= 2;
                            const uint32_t hf = Tr::mix(f, kmask), hr = Tr::mix(revcomp_code(f, k), kmask);
                            const uint32_t x = (hf < hr ? hf : hr) + 1;
                            const bool ok = i >= i_lo && i <= i_hi && !((uint32_t)bad & 1u) && x >= g;
                            bad >>= 1;
                            if (i < ic) streak = ok ? streak + 1 : 0;
                            else if (i > ic) {
                                alive = ok ? alive : 0u;
                                right += alive;
                            }
                        }
                        if ((int)(streak + right) >= w - 1) {
                            cnt = a.slot_rec[slot].y;
                            pos = (uint64_t)(gp - r0);
                        }
                    }
                }
            }
        }
        // ---- reserve output once per batch ----
        if (cnt) {
            my_off = atomicAdd(&s_cnt, cnt);
            atomicAdd(&s_nmin, 1u);
        }
        __syncthreads();
        // (atomics on one address retire at ~12 ns each device-wide: one per 1024 candidates, not one per wave)
        if (tid == 0 && s_cnt) {
            s_base = atomicAdd(a.n_hits, (unsigned long long)s_cnt);
            s_cnt = 0; // the next batch adds only after the barrier below
        }
        __syncthreads();
        if (cnt) {
            const unsigned long long at = s_base + my_off;
            if (at + cnt > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
                atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
            } else {
                const uint2 rec = a.slot_rec[slot];
                for (uint32_t qq = 0; qq < rec.y; ++qq) {
                    const uint32_t kn = a.rec_knode[rec.x + qq];
                    const uint32_t prg = a.rec_prg[rec.x + qq];
                    const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                    a.hit_key[at + qq] = pack_hit_key(read, prg, rev, (uint32_t)pos);
                    a.hit_val[at + qq] = kn >> 1;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0 && s_nmin) atomicAdd(a.n_minimizers, (unsigned long long)s_nmin);
}

uint32_t filter_n_tiles(uint64_t n_bases) { return (uint32_t)((n_bases + FT_WPOS - 1) / FT_WPOS); }

uint32_t filter_grid(uint32_t bloom_wbits, int n_cus, uint32_t n_tiles)
{
    // persistent grid: the two workgroups per CU that stay resident (64 KB of LDS and 1024 threads each), never more
    // waves than there are wave tiles
    (void)bloom_wbits;
    uint32_t grid = (uint32_t)n_cus * 2;
    const uint32_t need = (n_tiles + FT_WAVES - 1) / FT_WAVES;
    if (grid > need) grid = need;
    if (grid > (uint32_t)EX_MAX_WG) grid = EX_MAX_WG; // verify_expand_kernel keeps one prefix entry per workgroup in LDS
    return grid ? grid : 1;
}

hipError_t launch_sketch_filter(const SketchArgs& a, const uint32_t* bloom, uint32_t bloom_wbits, int n_cus, uint64_t* raw_pos,
    uint64_t raw_capacity, uint32_t* raw_count, hipStream_t stream, KernelTimer timer)
{
    if (a.n_bases == 0) return hipSuccess;
    if ((1u << bloom_wbits) > (uint32_t)FT_BLOOM_WORDS) return hipErrorInvalidValue;
    const uint32_t n_tiles = filter_n_tiles(a.n_bases);
    const size_t dyn = 16; // the candidate cursor, behind the static filter
    const bool short_k = a.k < 12;
    auto kernel = short_k ? &sketch_filter_kernel<true> : &sketch_filter_kernel<false>;
    static bool configured[2] = { false, false };
    if (!configured[short_k]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        configured[short_k] = true;
    }
    const uint32_t grid = filter_grid(bloom_wbits, n_cus, n_tiles);
    FilterArgs fa {};
    fa.bloom = bloom;
    fa.bloom_wbits = bloom_wbits;
    fa.n_tiles = n_tiles;
    fa.raw_pos = raw_pos;
    fa.raw_count = raw_count;
    fa.raw_slice = (uint32_t)std::min<uint64_t>(raw_capacity / grid, 0x7FFFFFFFull);
    if (const char* dbg = std::getenv("DRPRG_FT_DEBUG")) fa.debug = (uint32_t)std::atoi(dbg);
    if (timer.begin) HIP_TRY(hipEventRecord(timer.begin, stream));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(FT_THREADS), dyn, stream, a, fa);
    HIP_TRY(hipGetLastError());
    if (timer.end) HIP_TRY(hipEventRecord(timer.end, stream));
    hipLaunchKernelGGL(verify_expand_kernel, dim3((uint32_t)n_cus * 2), dim3(EX_THREADS), 0, stream, a, fa, grid);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
