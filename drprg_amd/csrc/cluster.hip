// cluster.hip -- K3: per-read hit clustering, size / overlap filters and atomic coverage accumulation on the
// sorted hit list (pandora define_clusters / filter_clusters / add_hits_to_kmergraphs).
#include "device_common.h"
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace drprg {
namespace dev {

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
// Only a few PRGs receive all the clusters, so per-cluster global atomics would serialise on a handful of
// addresses: histogram in LDS, then one global atomic per touched PRG and per counter per workgroup.
__global__ __launch_bounds__(256) void cluster_count_kernel(ClusterArgs a, uint32_t n_prgs)
{
    __shared__ uint32_t s_prg[MAX_PRGS];
    __shared__ uint32_t s_kept;
    __shared__ unsigned long long s_hits;
    const uint32_t n_clusters = *a.d_n_clusters;
    const uint32_t per_wg = (n_clusters + gridDim.x - 1) / gridDim.x;
    const uint32_t begin = blockIdx.x * per_wg;
    const uint32_t end = begin + per_wg < n_clusters ? begin + per_wg : n_clusters;
    if (begin >= end) return;
    for (uint32_t i = threadIdx.x; i < n_prgs; i += blockDim.x) s_prg[i] = 0;
    if (threadIdx.x == 0) { s_kept = 0; s_hits = 0; }
    __syncthreads();
    for (uint32_t c = begin + threadIdx.x; c < end; c += blockDim.x) {
        const ClusterRec r = a.clusters[c];
        if (r.state == 0) continue;
        atomicAdd(&s_prg[r.prg_rev >> 1], 1u);
        atomicAdd(&s_kept, 1u);
        atomicAdd(&s_hits, (unsigned long long)r.n);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_prgs; i += blockDim.x)
        if (s_prg[i]) atomicAdd(&a.prg_reads[i], s_prg[i]);
    if (threadIdx.x == 0 && s_kept) {
        atomicAdd(a.n_clusters_kept, (unsigned long long)s_kept);
        atomicAdd(a.n_hits_kept, s_hits);
    }
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

hipError_t exclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t stream)
{
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), stream);
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

hipError_t launch_cluster_pipeline(const ClusterArgs& a, uint32_t n_hits, uint32_t n_prgs, hipStream_t stream)
{
    const int B = 128;
    dim3 gc((n_hits + B - 1) / B); // n_clusters <= n_hits; the true count is read on the device
    hipLaunchKernelGGL(cluster_eval_kernel, gc, dim3(B), 0, stream, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(cluster_filter_kernel, gc, dim3(B), 0, stream, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(cluster_count_kernel, dim3(256), dim3(256), 0, stream, a, n_prgs);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(accumulate_kernel, dim3((n_hits + 255) / 256), dim3(256), 0, stream, a, n_hits);
    return hipGetLastError();
}

// dst[i] += src[i]: the device-side sum of two coverage vectors (several devices in one process: capi.cpp drprg_hip_reduce)
__global__ __launch_bounds__(256) void vector_add_u32_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] += src[i];
}

hipError_t launch_vector_add_u32(uint32_t* dst, const uint32_t* src, uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(vector_add_u32_kernel, dim3(grid), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
