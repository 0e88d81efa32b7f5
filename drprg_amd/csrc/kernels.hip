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
    static constexpr uint32_t INVALID = 0xFFFFFFFFu;
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
    static constexpr uint64_t INVALID = ~0ULL;
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
__device__ inline uint32_t encode4(uint32_t word)
{
    return encode_base(word & 0xFF) | (encode_base((word >> 8) & 0xFF) << 8) | (encode_base((word >> 16) & 0xFF) << 16)
        | (encode_base(word >> 24) << 24);
}

// first read r with offsets[r+1] > gp, i.e. the read holding global base position gp
__device__ inline uint32_t find_read(const uint64_t* __restrict__ offsets, uint32_t n_reads, uint64_t gp)
{
    uint32_t lo = 0, hi = n_reads; // invariant: offsets[lo] <= gp < offsets[hi]
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
constexpr int SK_MAXTAIL = 32;                 // k-1 <= 31 extra bases
constexpr int SK_CODES = SK_NPOS + 48;         // staged bases (multiple of 16 >= NPOS + k - 1)

template <typename HT>
__global__ __launch_bounds__(SK_THREADS) void sketch_probe_kernel(SketchArgs a)
{
    using Tr = HashTraits<HT>;
    __shared__ __attribute__((aligned(16))) uint8_t s_code[SK_CODES];
    __shared__ HT s_hash[SK_NPOS];
    __shared__ uint16_t s_strand[SK_THREADS];
    __shared__ uint16_t s_mins[SK_NPOS];
    __shared__ uint32_t s_nmin;
    __shared__ uint32_t s_first_read;

    const int tid = threadIdx.x;
    const int w = a.w, k = a.k;
    const int halo = a.halo;                 // multiple of 16, >= w-1
    const int t_eval = SK_NPOS - 2 * halo;   // k-mer positions evaluated by this tile
    // tile origin in global base coordinates (may be negative for tile 0)
    const int64_t origin = (int64_t)blockIdx.x * t_eval - halo;
    const int64_t n_bases = (int64_t)a.n_bases;

    if (tid == 0) s_nmin = 0;

    // ---- stage bases -> codes (coalesced 16-byte loads; origin is a multiple of 16) ----
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
        *reinterpret_cast<uint4*>(&s_code[v * 16]) = out;
    }
    // ---- locate the first read starting inside the staged range ----
    if (tid == 0) {
        int64_t lo_pos = origin < 0 ? 0 : origin;
        // first r with offsets[r] >= lo_pos
        uint32_t lo = 0, hi = a.n_reads; // offsets[n_reads] = n_bases >= anything staged
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)a.offsets[mid] < lo_pos) lo = mid + 1; else hi = mid;
        }
        s_first_read = lo;
    }
    __syncthreads();
    // ---- flag read starts (bit 3) ----
    {
        const int64_t end_pos = origin + SK_CODES;
        for (uint32_t r = s_first_read + tid; r < a.n_reads; r += SK_THREADS) {
            int64_t o = (int64_t)a.offsets[r];
            if (o >= end_pos) break;
            s_code[o - origin] |= 8;
        }
    }
    __syncthreads();

    // ---- phase 1: rolling canonical hash of SK_G consecutive k-mers per thread ----
    {
        const HT mask = (HT)((k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1));
        const int shift1 = 2 * (k - 1);
        HT fwd = 0, rev = 0;
        int run = 0;
        uint32_t strand_bits = 0;
        const int base0 = tid * SK_G;
        const int nsteps = SK_G + k - 1;
        for (int s0 = 0; s0 < nsteps; s0 += 4) {
            uint32_t word = *reinterpret_cast<const uint32_t*>(&s_code[base0 + s0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s0 + q;
                uint32_t c = (word >> (8 * q)) & 0xFF;
                uint32_t b = c & 3;
                fwd = ((fwd << 2) | (HT)b) & mask;
                rev = (rev >> 2) | ((HT)(3 - b) << shift1);
                run = (c & 4) ? 0 : ((c & 8) ? 1 : run + 1);
                if (s >= k - 1 && s < nsteps) {
                    int j = s - (k - 1); // k-mer index within this thread
                    HT h = Tr::INVALID;
                    if (run >= k) {
                        HT hf = Tr::mix(fwd, mask), hr = Tr::mix(rev, mask);
                        h = hf < hr ? hf : hr;
                        strand_bits |= (uint32_t)(hf <= hr) << j;
                    }
                    s_hash[base0 + j] = h;
                }
            }
        }
        s_strand[tid] = (uint16_t)strand_bits;
    }
    __syncthreads();

    // ---- phase 2a: which of my positions are window minimizers? ----
    // position j is a minimizer iff a window of w valid k-mers containing j has no hash below h[j]:
    // count neighbours >= h[j] leftwards (a) and rightwards (b) until w-1 are found.
    {
        const int lo = halo, hi = SK_NPOS - halo;
        const int64_t last_kmer = n_bases - k; // last valid global k-mer start
        for (int g = 0; g < SK_G; ++g) {
            int j = tid * SK_G + g;
            if (j < lo || j >= hi) continue;
            if (origin + j > last_kmer) continue;
            HT h = s_hash[j];
            if (h == Tr::INVALID) continue;
            int need = w - 1, got = 0;
            for (int d = 1; d <= need; ++d) { // leftwards (j-d >= 0 because halo >= w-1)
                HT x = s_hash[j - d];
                if (x == Tr::INVALID || x < h) break;
                ++got;
            }
            for (int d = 1; got < need && d <= need; ++d) {
                HT x = s_hash[j + d];
                if (x == Tr::INVALID || x < h) break;
                ++got;
            }
            if (got >= need) {
                uint32_t idx = atomicAdd(&s_nmin, 1u);
                s_mins[idx] = (uint16_t)j;
            }
        }
    }
    __syncthreads();

    // ---- phase 2b: probe the index with the compacted minimizer list ----
    const uint32_t nmin = s_nmin;
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const HT* __restrict__ slot_key = reinterpret_cast<const HT*>(a.slot_key);
    for (uint32_t i = tid; i < nmin; i += SK_THREADS) {
        int j = s_mins[i];
        HT h = s_hash[j];
        uint32_t s = table_slot_dev((uint64_t)h, a.table_bits);
        uint2 rec = make_uint2(0, 0);
        while (true) {
            uint2 r = a.slot_rec[s];
            if (r.y == 0) break;
            if (slot_key[s] == h) { rec = r; break; }
            s = (s + 1) & tmask;
        }
        if (rec.y == 0) continue;
        // a hit: locate the read and emit one hit per index record
        uint64_t gp = (uint64_t)(origin + j);
        uint32_t read = find_read(a.offsets, a.n_reads, gp);
        uint64_t pos = gp - a.offsets[read];
        uint32_t strand = (s_strand[j / SK_G] >> (j % SK_G)) & 1u;
        unsigned long long at = atomicAdd(a.n_hits, (unsigned long long)rec.y);
        if (at + rec.y > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
            atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
            continue;
        }
        for (uint32_t q = 0; q < rec.y; ++q) {
            uint32_t kn = a.rec_knode[rec.x + q]; // (global knode << 1) | strand
            uint32_t prg = a.rec_prg[rec.x + q];
            uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u; // forward hits sort first
            a.hit_key[at + q] = pack_hit_key(read, prg, rev, (uint32_t)pos);
            a.hit_val[at + q] = kn >> 1;
        }
    }
    if (tid == 0 && nmin) atomicAdd(a.n_minimizers, (unsigned long long)nmin);
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

hipError_t launch_sketch_probe(const SketchArgs& a, bool wide_hash, hipStream_t stream)
{
    if (a.n_bases == 0) return hipSuccess;
    uint64_t t_eval = sketch_tile_eval(a.halo);
    uint64_t n_kmer_pos = a.n_bases; // positions past n_bases-k are skipped inside the kernel
    uint32_t grid = (uint32_t)((n_kmer_pos + t_eval - 1) / t_eval);
    if (wide_hash)
        hipLaunchKernelGGL(sketch_probe_kernel<uint64_t>, dim3(grid), dim3(SK_THREADS), 0, stream, a);
    else
        hipLaunchKernelGGL(sketch_probe_kernel<uint32_t>, dim3(grid), dim3(SK_THREADS), 0, stream, a);
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
