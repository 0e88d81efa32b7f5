// sketch_filter.hip -- K1+K2 in their fast form: persistent workgroups + an LDS-resident Bloom prefilter.
//
// A read minimizer can only produce a hit if its k-mer is an index k-mer (in either orientation).  So instead of
// hashing every k-mer of every read (sketch_probe.hip: ~86 VALU instructions per base, two 15-op hashes each),
// this kernel tests each position's 2-bit k-mer *code* against a Bloom filter of the index k-mer codes that stays
// in LDS for the lifetime of a persistent workgroup.  Only candidates get the exact treatment -- canonical hashes
// of the 2w-1 neighbouring k-mers and the window-minimizer test -- and the survivors are written, without any
// global atomic, to a per-workgroup slice of a raw-hit buffer.  A second, fully parallel kernel
// (expand_hits_kernel) does the exact table lookup, finds the read of each raw hit and emits the (key,val) hits
// the cluster pipeline sorts.  The Bloom filter has no false negatives, so the result is identical to the direct
// kernel (tests/test_gpu_parity.py checks both against the oracle).  Serves k <= 15, w <= 16 and indexes whose
// filter fits 64 KB of LDS; everything else takes the direct kernel.
//
// Nothing in the per-tile loop waits on global memory: the next tile's bases, its first-read index and its read
// offsets are prefetched into registers while the current tile is processed from LDS.
#include "device_common.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace drprg {
namespace dev {

constexpr int FT_THREADS = 512;
constexpr int FT_G = 16;
constexpr int FT_NPOS = FT_THREADS * FT_G;   // 8192 positions per tile
constexpr int FT_HALO = 16;                  // >= w-1
constexpr int FT_EVAL = FT_NPOS - 2 * FT_HALO;
constexpr int FT_CODES = FT_NPOS + 48;       // staged bases
constexpr int FT_WORDS = FT_CODES / 16;      // 515 packed words
constexpr int FT_CAND_CAP = 2 * FT_THREADS;  // up to two candidates per thread per round
constexpr int FT_VER_HITS = 32;              // candidates verified per pass
constexpr int FT_VER_W = 31;                 // 2*16-1 neighbour slots
constexpr int FT_START_WORDS = (FT_CODES + 31) / 32 + 1;

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
    // number of reads that start inside the staged range of the tile (so that exactly those offsets are loaded)
    const int64_t end_pos = (int64_t)b * FT_EVAL - FT_HALO + FT_CODES;
    uint32_t lo2 = lo, hi2 = n_reads;
    while (lo2 < hi2) {
        uint32_t mid = lo2 + ((hi2 - lo2) >> 1);
        if ((int64_t)offsets[mid] < end_pos) lo2 = mid + 1; else hi2 = mid;
    }
    out[n_tiles + b] = lo2 - lo;
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
    uint32_t start[FT_START_WORDS]; // bit per staged base: a read starts here
    uint16_t cand[FT_CAND_CAP];
    uint32_t ncand;
    uint32_t more; // some thread still holds candidates for another round
    uint32_t nraw; // raw hits appended by this workgroup so far (may exceed its slice: overflow)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would stall every wave
// on the next tile's prefetch (global loads still in flight on purpose); LDS operations of a wave complete in
// order, so lgkmcnt(0) + s_barrier is enough for LDS visibility inside the workgroup.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FilterArgs {
    const uint32_t* bloom;
    uint32_t bloom_wbits;
    uint32_t n_tiles;
    uint64_t* raw_pos;   // [grid][raw_slice]: global base position | strand << 63
    uint32_t* raw_hash;  // canonical hash of the minimizer
    uint32_t* raw_hint;  // a read at or before the one holding the position (start of the read search)
    uint32_t* raw_count; // [grid]
    uint32_t raw_slice;
    uint32_t debug; // ablation switches for profiling (DRPRG_FT_DEBUG): 1 = skip the Bloom test, 2 = skip verification
};

__global__ __launch_bounds__(FT_THREADS) void sketch_filter_kernel(SketchArgs a, FilterArgs fa)
{
    using Tr = HashTraits<uint32_t>;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bloom[];
    __shared__ FtShared sh;

    const int tid = threadIdx.x;
    const int k = a.k, w = a.w;
    const uint32_t kmask = (1u << (2 * k)) - 1;
    const uint32_t kbits = (1u << k) - 1; // k consecutive base flags
    const int64_t n_bases = (int64_t)a.n_bases;
    const int base0 = tid * FT_G;
    const uint32_t n_tiles = fa.n_tiles;

    for (uint32_t i = tid; i < (1u << fa.bloom_wbits); i += FT_THREADS) s_bloom[i] = fa.bloom[i];
    if (tid == 0) sh.nraw = 0;

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
    // k-mer at tile position p: canonical hash + 1, or 0 if it holds an N or straddles two reads
    auto kmer_at = [&](int p, bool& strand) -> uint32_t {
        const int v = p >> 4, o = p & 15;
        const uint32_t f = __funnelshift_l(sh.pack[v + 1], sh.pack[v], 2 * o) >> (32 - 2 * k);
        const uint32_t nm = (uint32_t)sh.nmask[v] | ((uint32_t)sh.nmask[v + 1] << 16);
        if ((nm >> o) & kbits) return 0;
        const int q = p + 1; // a read starting at p+1 .. p+k-1 splits the k-mer
        const uint32_t st = __funnelshift_r(sh.start[q >> 5], sh.start[(q >> 5) + 1], q & 31);
        if (st & (kbits >> 1)) return 0;
        const uint32_t hf = Tr::mix(f, kmask), hr = Tr::mix(revcomp_code(f, k), kmask);
        strand = hf <= hr;
        return (hf < hr ? hf : hr) + 1;
    };

    unsigned long long dbg_cands = 0;
    // register ring: the tile being processed plus three tiles in flight (Little's law: ~48 KB per CU must be
    // outstanding to keep HBM busy; one tile is 8 KB per workgroup)
    struct Pending {
        uint4 main, extra;
        int64_t off;
        uint2 fc; // first read starting in the tile's staged range, number of reads starting there
    };
    const uint32_t gs = gridDim.x; // tile stride of this persistent workgroup
    auto fetch = [&](uint32_t t, uint2 fc, Pending& p) {
        p.fc = fc;
        p.off = INT64_MAX;
        if (t < n_tiles) {
            load_tile(t, p.main, p.extra);
            if ((uint32_t)tid < fc.y) p.off = (int64_t)a.offsets[(uint64_t)fc.x + tid];
        }
    };
    auto first_of = [&](uint32_t t) -> uint2 {
        return t < n_tiles ? make_uint2(a.tile_first_read[t], a.tile_first_read[n_tiles + t]) : make_uint2(0u, 0u);
    };
    Pending cur {}, p1 {}, p2 {}, p3 {};
    uint32_t tile = blockIdx.x;
    fetch(tile, first_of(tile), cur);
    fetch(tile + gs, first_of(tile + gs), p1);
    fetch(tile + 2 * gs, first_of(tile + 2 * gs), p2);
    uint2 first3 = first_of(tile + 3 * gs);
    __syncthreads(); // Bloom filter in place

    for (; tile < n_tiles; tile += gs) {
        const int64_t origin = (int64_t)tile * FT_EVAL - FT_HALO;
        const uint32_t first_read = cur.fc.x, n_starts = cur.fc.y;
        const int64_t cur_off = cur.off;
        const uint2 first4 = first_of(tile + 4 * gs); // scalar loads, consumed in the next iteration

        // ---- pack this tile into LDS ----
        {
            uint32_t pk, nm;
            pack16(cur.main, pk, nm);
            sh.pack[tid] = pk;
            sh.nmask[tid] = (uint16_t)nm;
            if (tid < FT_WORDS - FT_THREADS) {
                pack16(cur.extra, pk, nm);
                sh.pack[FT_THREADS + tid] = pk;
                sh.nmask[FT_THREADS + tid] = (uint16_t)nm;
            }
            for (int i = tid; i < FT_START_WORDS; i += FT_THREADS) sh.start[i] = 0;
            if (tid == 0) {
                sh.pack[FT_WORDS] = 0;
                sh.nmask[FT_WORDS] = 0xFFFF;
            }
        }
        lds_barrier(); // not __syncthreads(): the ring's global loads must stay in flight
        // ---- read starts of this tile -> bitmap (first offset per thread was prefetched) ----
        {
            for (uint32_t i = tid; i < n_starts; i += FT_THREADS) {
                // beyond the first 512 starts (very short reads only) the offsets are loaded here
                const int64_t o = i < FT_THREADS ? cur_off : (int64_t)a.offsets[(uint64_t)first_read + i];
                const int oc = (int)(o - origin);
                atomicOr(&sh.start[oc >> 5], 1u << (oc & 31));
            }
        }
        // ---- prefetch: tile + 3*grid joins the ring; its loads stay in flight while tiles are processed from LDS
        // (issued after this tile's own offset loads, because vector-memory results return in issue order) ----
        fetch(tile + 3 * gs, first3, p3);
        // ---- Bloom test of my 16 positions ----
        uint32_t cand = 0;
        if (!(fa.debug & 1u) && base0 >= FT_HALO && base0 < FT_NPOS - FT_HALO) {
            const uint32_t w0 = sh.pack[tid], w1 = sh.pack[tid + 1];
            const int sh_k = 32 - 2 * k, sh_w = 32 - (int)fa.bloom_wbits;
#pragma unroll
            for (int j = 0; j < FT_G; ++j) {
                const uint32_t f = __funnelshift_l(w1, w0, 2 * j) >> sh_k;
                const uint32_t hsh = f * 0x9E3779B1u;
                const uint32_t word = s_bloom[hsh >> sh_w];
                cand |= ((word >> (hsh & 31)) & (word >> ((hsh >> 5) & 31)) & (word >> ((hsh >> 10) & 31)) & 1u) << j;
            }
        }
        // ---- rounds (almost always one): compact candidates, window test from LDS, append raw hits ----
        const int span = 2 * w - 1;
        if (fa.debug & 2u) cand = 0;
        while (true) {
            if (tid == 0) {
                sh.ncand = 0;
                sh.more = 0;
            }
            lds_barrier(); // also orders the start bitmap before its first use
            if (cand) {
                int np = __popc(cand);
                if (np > 2) np = 2;
                uint32_t at = atomicAdd(&sh.ncand, (uint32_t)np);
                for (int i = 0; i < np; ++i) {
                    const int j = __ffs(cand) - 1;
                    cand &= cand - 1;
                    sh.cand[at++] = (uint16_t)(base0 + j);
                }
            }
            if (cand) sh.more = 1;
            lds_barrier();
            const uint32_t more = sh.more;
            const uint32_t ncand = sh.ncand;
            dbg_cands += ncand;
            // Window test, half a wave per candidate: lane d of the half evaluates neighbour d (d = w-1 is the candidate
            // itself); a ballot of "valid and >= the candidate's hash" gives the run of such neighbours on either side.
            {
                const int lane = tid & 63, half = lane >> 5, d = lane & 31;
                for (uint32_t cb = (uint32_t)(tid >> 6) * 2; cb < ncand; cb += (FT_THREADS / 64) * 2) {
                    const uint32_t c = cb + (uint32_t)half;
                    const bool active = c < ncand && d < span;
                    const int p = c < ncand ? (int)sh.cand[c] : FT_HALO;
                    bool strand = false;
                    const uint32_t g = active ? kmer_at(p + d - (w - 1), strand) : 0u;
                    const uint32_t gc = (uint32_t)__shfl((int)g, half * 32 + (w - 1));
                    const uint64_t ball = __ballot(active && g != 0 && g >= gc);
                    const uint32_t m = half ? (uint32_t)(ball >> 32) : (uint32_t)ball;
                    const uint32_t lmask = (1u << (w - 1)) - 1;       // bits 0 .. w-2: left neighbours, bit w-2 nearest
                    const uint32_t lzero = ~m & lmask;
                    const int left = lzero ? (w - 2) - (31 - __clz((int)lzero)) : (w - 1);
                    const uint32_t rzero = ~(m >> w);                 // bit 0: nearest right neighbour
                    int right = __ffs((int)rzero) - 1;
                    if (right > w - 1) right = w - 1;
                    // a window of w valid k-mers around p without a smaller hash exists: p is a read minimizer
                    if (d == w - 1 && c < ncand && gc != 0 && left + right >= w - 1) {
                        const uint32_t idx = atomicAdd(&sh.nraw, 1u);
                        if (idx < fa.raw_slice) {
                            const size_t at = (size_t)blockIdx.x * fa.raw_slice + idx;
                            fa.raw_pos[at] = (uint64_t)(origin + p) | ((uint64_t)strand << 63);
                            fa.raw_hash[at] = g - 1;
                            fa.raw_hint[at] = first_read ? first_read - 1 : 0;
                        }
                    }
                }
            }
            lds_barrier(); // the candidate list is reused by the next round
            if (!more) break;
        }
        lds_barrier(); // everyone is done with this tile's LDS before it is overwritten
        cur = p1;
        p1 = p2;
        p2 = p3;
        first3 = first4;
    }
    if (tid == 0) {
        fa.raw_count[blockIdx.x] = sh.nraw;
        if (fa.debug & 4u) atomicAdd(a.n_minimizers, dbg_cands | ((unsigned long long)sh.nraw << 40));
        if (sh.nraw > fa.raw_slice) atomicOr(a.overflow, 4u);
    }
}

// raw hits -> hits: exact table lookup, read lookup, one (key,val) per index record.
// Every workgroup takes a contiguous range of the concatenated raw slices; output space is reserved once per
// 256-thread batch (LDS prefix + one global atomic per workgroup per batch), not once per wave.
constexpr int EX_MAX_WG = 1024;
__global__ __launch_bounds__(256) void expand_hits_kernel(SketchArgs a, FilterArgs fa, uint32_t n_wg)
{
    using Tr = HashTraits<uint32_t>;
    __shared__ uint32_t s_prefix[EX_MAX_WG + 1];
    __shared__ uint32_t s_cnt, s_found;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    // inclusive scan of the (clamped) slice counts: 4 entries per thread, then a block scan
    {
        uint32_t v[4], run = 0;
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = (uint32_t)tid * 4 + i;
            const uint32_t n = b < n_wg ? fa.raw_count[b] : 0u;
            v[i] = run;
            run += n < fa.raw_slice ? n : fa.raw_slice;
        }
        __shared__ uint32_t s_part[256];
        s_part[tid] = run;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t add = tid >= off ? s_part[tid - off] : 0u;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const uint32_t before = tid ? s_part[tid - 1] : 0u;
        for (int i = 0; i < 4; ++i) s_prefix[tid * 4 + i] = before + v[i];
        if (tid == 255) s_prefix[EX_MAX_WG] = s_part[255];
        __syncthreads();
    }
    const uint32_t total = s_prefix[EX_MAX_WG];
    const uint32_t* __restrict__ slot_key = reinterpret_cast<const uint32_t*>(a.slot_key);
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const uint32_t per_wg = (total + gridDim.x - 1) / gridDim.x;
    const uint32_t t_begin = blockIdx.x * per_wg;
    const uint32_t t_end = t_begin + per_wg < total ? t_begin + per_wg : total;
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += 256) {
        const uint32_t t = t0 + tid;
        uint32_t cnt = 0, slot = 0, strand = 0, hint = 0;
        uint64_t gp = 0;
        if (t < t_end) {
            uint32_t lo = 0, hi = EX_MAX_WG; // s_prefix[lo] <= t < s_prefix[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= t) lo = mid; else hi = mid;
            }
            const size_t at_raw = (size_t)lo * fa.raw_slice + (t - s_prefix[lo]);
            const uint64_t rp = fa.raw_pos[at_raw];
            strand = (uint32_t)(rp >> 63);
            gp = rp & ~(1ull << 63);
            hint = fa.raw_hint[at_raw];
            const uint32_t h = fa.raw_hash[at_raw];
            uint32_t s = table_slot_dev((uint64_t)h, a.table_bits);
            while (true) {
                const uint32_t key = slot_key[s];
                if (key == h) { slot = s; cnt = a.slot_rec[s].y; break; }
                if (key == Tr::EMPTY) break; // a Bloom false positive
                s = (s + 1) & tmask;
            }
        }
        if (tid == 0) { s_cnt = 0; s_found = 0; }
        __syncthreads();
        uint32_t my_off = 0;
        if (cnt) {
            my_off = atomicAdd(&s_cnt, cnt);
            atomicAdd(&s_found, 1u);
        }
        __syncthreads();
        if (tid == 0 && s_cnt) {
            s_base = atomicAdd(a.n_hits, (unsigned long long)s_cnt);
            atomicAdd(a.n_minimizers, (unsigned long long)s_found);
        }
        __syncthreads();
        if (cnt) {
            const uint32_t read = find_read_from(a.offsets, a.n_reads, hint, gp);
            const uint64_t pos = gp - a.offsets[read];
            const unsigned long long at = s_base + my_off;
            if (at + cnt > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
                atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
            } else {
                const uint2 rec = a.slot_rec[slot];
                for (uint32_t q = 0; q < rec.y; ++q) {
                    const uint32_t kn = a.rec_knode[rec.x + q];
                    const uint32_t prg = a.rec_prg[rec.x + q];
                    const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                    a.hit_key[at + q] = pack_hit_key(read, prg, rev, (uint32_t)pos);
                    a.hit_val[at + q] = kn >> 1;
                }
            }
        }
        __syncthreads();
    }
}

uint32_t filter_n_tiles(uint64_t n_bases) { return (uint32_t)((n_bases + FT_EVAL - 1) / FT_EVAL); }

uint32_t filter_grid(uint32_t bloom_wbits, int n_cus, uint32_t n_tiles)
{
    // persistent grid: as many workgroups as stay resident (LDS-limited), never more than there are tiles
    // (LDS is handed out in granules; leave a margin so that the resident count is not over-estimated)
    const size_t lds_per_wg = (sizeof(uint32_t) << bloom_wbits) + sizeof(FtShared) + 2048;
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds_per_wg);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    uint32_t grid = (uint32_t)n_cus * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid > (uint32_t)EX_MAX_WG) grid = EX_MAX_WG; // expand_hits_kernel keeps one prefix entry per workgroup in LDS
    return grid ? grid : 1;
}

hipError_t launch_sketch_filter(const SketchArgs& a, const uint32_t* bloom, uint32_t bloom_wbits, int n_cus, uint64_t* raw_pos,
    uint32_t* raw_hash, uint32_t* raw_hint, uint64_t raw_capacity, uint32_t* raw_count, hipStream_t stream, KernelTimer timer)
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
    const uint32_t grid = filter_grid(bloom_wbits, n_cus, n_tiles);
    FilterArgs fa {};
    fa.bloom = bloom;
    fa.bloom_wbits = bloom_wbits;
    fa.n_tiles = n_tiles;
    fa.raw_pos = raw_pos;
    fa.raw_hash = raw_hash;
    fa.raw_hint = raw_hint;
    fa.raw_count = raw_count;
    fa.raw_slice = (uint32_t)std::min<uint64_t>(raw_capacity / grid, 0x7FFFFFFFull);
    if (const char* dbg = std::getenv("DRPRG_FT_DEBUG")) fa.debug = (uint32_t)std::atoi(dbg);
    if (timer.begin) HIP_TRY(hipEventRecord(timer.begin, stream));
    hipLaunchKernelGGL(sketch_filter_kernel, dim3(grid), dim3(FT_THREADS), dyn, stream, a, fa);
    HIP_TRY(hipGetLastError());
    if (timer.end) HIP_TRY(hipEventRecord(timer.end, stream));
    hipLaunchKernelGGL(expand_hits_kernel, dim3((uint32_t)n_cus * 8), dim3(256), 0, stream, a, fa, grid);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
