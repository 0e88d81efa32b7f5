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
//   verify_expand_kernel   one thread per candidate, no barriers on the critical path: canonical hash from the raw
//                          bases -> exact table lookup (false positives end here) -> read lookup -> window-minimizer
//                          test over the 2w-1 neighbouring k-mers inside the read -> one (key,val) hit per index
//                          record, with output space reserved once per 256-thread batch.
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

// canonical hash + 1 of the k-mer starting at bases[p] (k <= 15), 0 if it holds a non-ACGT base.  Two aligned 16-byte
// loads cover any 15-mer; the byte loop only serves the last bytes of the buffer.
__device__ inline uint32_t kmer_hash_at(const uint8_t* __restrict__ bases, int64_t n_bases, int64_t p, int k, uint32_t kmask,
    bool& strand)
{
    uint32_t f;
    const int64_t a0 = p & ~(int64_t)15;
    if (a0 + 32 <= n_bases) {
        const uint4 A = *reinterpret_cast<const uint4*>(bases + a0), B = *reinterpret_cast<const uint4*>(bases + a0 + 16);
        uint32_t pa, pb, na, nb;
        pack16n(A, pa, na);
        pack16n(B, pb, nb);
        const int o = (int)(p - a0);
        if (((na | (nb << 16)) >> o) & ((1u << k) - 1)) return 0;
        f = __funnelshift_l(pb, pa, 2 * o) >> (32 - 2 * k);
    } else {
        uint32_t bad = 0;
        f = 0;
        for (int i = 0; i < k; ++i) {
            const uint32_t c = encode_base(bases[p + i]);
            bad |= c;
            f = (f << 2) | (c & 3u);
        }
        if (bad & 4u) return 0;
    }
    const uint32_t hf = HashTraits<uint32_t>::mix(f, kmask), hr = HashTraits<uint32_t>::mix(revcomp_code(f, k), kmask);
    strand = hf <= hr;
    return (hf < hr ? hf : hr) + 1;
}

constexpr int EX_MAX_WG = 1024;
constexpr int EX_THREADS = 256;

struct ExFound { // a candidate whose canonical hash is an index key
    int64_t gp, r0, r1; // position, start and end of its read (global base coordinates)
    uint32_t slot, read, g, strand, cnt;
};

__global__ __launch_bounds__(EX_THREADS) void verify_expand_kernel(SketchArgs a, FilterArgs fa, uint32_t n_wg)
{
    using Tr = HashTraits<uint32_t>;
    __shared__ uint32_t s_prefix[EX_MAX_WG + 1];
    __shared__ uint32_t s_part[EX_THREADS];
    __shared__ ExFound s_found[EX_THREADS];
    __shared__ uint32_t s_nfound, s_cnt, s_nmin;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    // exclusive prefix sums of the (clamped) slice counts: 4 entries per thread, then a block scan
    {
        uint32_t v[4], run = 0;
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = (uint32_t)tid * 4 + i;
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
        for (int i = 0; i < 4; ++i) s_prefix[tid * 4 + i] = before + v[i];
        if (tid == EX_THREADS - 1) s_prefix[EX_MAX_WG] = s_part[EX_THREADS - 1];
        __syncthreads();
    }
    const uint32_t total = s_prefix[EX_MAX_WG];
    const uint32_t* __restrict__ slot_key = reinterpret_cast<const uint32_t*>(a.slot_key);
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const int k = a.k, w = a.w;
    const uint32_t kmask = (1u << (2 * k)) - 1;
    const int64_t n_bases = (int64_t)a.n_bases;
    const uint32_t per_wg = (total + gridDim.x - 1) / gridDim.x;
    const uint32_t t_begin = blockIdx.x * per_wg;
    const uint32_t t_end = t_begin + per_wg < total ? t_begin + per_wg : total;
    const int lane = tid & 63, half = lane >> 5, d = lane & 31;
    const double reads_per_base = (double)a.n_reads / (double)(a.n_bases ? a.n_bases : 1);
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += EX_THREADS) {
        if (tid == 0) { s_nfound = 0; s_cnt = 0; s_nmin = 0; }
        __syncthreads();
        // ---- phase 1, one lane per candidate: canonical hash from the raw bases, exact table lookup, read lookup ----
        const uint32_t t = t0 + tid;
        if (t < t_end) {
            uint32_t lo = 0, hi = EX_MAX_WG; // s_prefix[lo] <= t < s_prefix[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= t) lo = mid; else hi = mid;
            }
            const int64_t gp = (int64_t)fa.raw_pos[(size_t)lo * fa.raw_slice + (t - s_prefix[lo])];
            bool st = false;
            const uint32_t g = gp + k <= n_bases ? kmer_hash_at(a.bases, n_bases, gp, k, kmask, st) : 0u;
            if (g) { // Bloom false positives end at the exact lookup
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
                    // interpolated first guess: exact for fixed-length reads, a short gallop otherwise
                    const uint32_t guess = (uint32_t)((double)gp * reads_per_base);
                    const uint32_t rlo = find_read_near(a.offsets, a.n_reads, guess, (uint64_t)gp);
                    const int64_t r0 = (int64_t)a.offsets[rlo], r1 = (int64_t)a.offsets[rlo + 1];
                    if (gp + k <= r1) { // the k-mer lies inside one read
                        ExFound e;
                        e.gp = gp; e.r0 = r0; e.r1 = r1;
                        e.slot = s; e.read = rlo; e.g = g; e.strand = st ? 1u : 0u; e.cnt = 0;
                        s_found[atomicAdd(&s_nfound, 1u)] = e;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2, half a wave per index k-mer: lane d evaluates neighbour d - (w-1); a ballot of "inside the read,
        // no N, hash >= the candidate's" gives the runs on either side; a window of w such k-mers makes it a minimizer ----
        const uint32_t nfound = s_nfound;
        for (uint32_t cb = (uint32_t)(tid >> 6) * 2; cb < nfound; cb += (EX_THREADS / 64) * 2) {
            const uint32_t c = cb + (uint32_t)half;
            const bool have = c < nfound;
            const ExFound e = s_found[have ? c : 0];
            const int64_t q = e.gp + d - (w - 1);
            uint32_t x = 0;
            bool s2;
            if (have && d < 2 * w - 1 && d != w - 1 && q >= e.r0 && q + k <= e.r1) x = kmer_hash_at(a.bases, n_bases, q, k, kmask, s2);
            const uint64_t ball = __ballot(x != 0 && x >= e.g);
            const uint32_t m = half ? (uint32_t)(ball >> 32) : (uint32_t)ball;
            const uint32_t lmask = (1u << (w - 1)) - 1;   // bits 0 .. w-2: left neighbours, bit w-2 nearest
            const uint32_t lzero = ~m & lmask;
            const int left = lzero ? (w - 2) - (31 - __clz((int)lzero)) : (w - 1);
            const uint32_t rzero = ~(m >> w);             // bit 0: nearest right neighbour
            int right = __ffs((int)rzero) - 1;
            if (right > w - 1) right = w - 1;
            if (have && d == w - 1 && left + right >= w - 1) {
                const uint32_t cnt = a.slot_rec[e.slot].y;
                s_found[c].cnt = cnt;
                atomicAdd(&s_cnt, cnt);
                atomicAdd(&s_nmin, 1u);
            }
        }
        __syncthreads();
        // ---- phase 3: reserve output once per batch, emit one (key,val) per index record ----
        if (tid == 0 && s_cnt) {
            s_base = atomicAdd(a.n_hits, (unsigned long long)s_cnt);
            atomicAdd(a.n_minimizers, (unsigned long long)s_nmin);
            s_cnt = 0; // reused as the running offset below
        }
        __syncthreads();
        if ((uint32_t)tid < nfound && s_found[tid].cnt) {
            const ExFound e = s_found[tid];
            const unsigned long long at = s_base + atomicAdd(&s_cnt, e.cnt);
            const uint64_t pos = (uint64_t)(e.gp - e.r0);
            if (at + e.cnt > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
                atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
            } else {
                const uint2 rec = a.slot_rec[e.slot];
                for (uint32_t qq = 0; qq < rec.y; ++qq) {
                    const uint32_t kn = a.rec_knode[rec.x + qq];
                    const uint32_t prg = a.rec_prg[rec.x + qq];
                    const uint32_t rev = ((kn & 1u) == e.strand) ? 0u : 1u;
                    a.hit_key[at + qq] = pack_hit_key(e.read, prg, rev, (uint32_t)pos);
                    a.hit_val[at + qq] = kn >> 1;
                }
            }
        }
        __syncthreads();
    }
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
    hipLaunchKernelGGL(verify_expand_kernel, dim3((uint32_t)n_cus * 8), dim3(256), 0, stream, a, fa, grid);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
