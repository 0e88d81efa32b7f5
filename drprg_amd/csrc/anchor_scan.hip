// anchor_scan.hip -- which of the reads that are resident in HBM hold an anchor k-mer of a candidate region?
//
// `pandora discover` (spawned at /root/reference/src/predict.rs:248-256) reads the sample a second time to assemble the candidate
// regions; the pile-up of this build (denovo.cpp) needs only the reads that hold the exact anchor k-mers on both sides of a
// region -- a few thousand of ten million.  When the mapping pass left the batches in HBM (Mapper::keep_reads) this kernel finds
// them there: one pass over the base stream at HBM speed instead of a second pass over the file on the host (10 M x 150 bp: ~1 ms
// against ~200 ms of 32 parser threads), and the host pile-up then runs unchanged on the reads it selected.
//
// The selection is a superset by construction: a read is taken as soon as ONE k-mer that starts inside it equals an anchor (the host
// asks for two anchors, left and right, of one region), including k-mers that run over the end of the read into the next one of the
// stream -- the host scan that follows decides, with the code that scans a file.
//
// Work: the base stream of a batch is one array; a thread owns 32 consecutive k-mer END positions and reads the 32 bases before
// them as well (anchors are at most 31 bases), as four aligned 16-byte loads; a wave covers 2 KB of the stream per step.  The k-mer
// is rolled 2 bits per base, A=0 C=1 G=2 T=3 (nt4 of common.h, either case), a base that is none of them restarts the run.  Low 16
// bits of every full k-mer -> one bit of an 8 KB table in LDS; the rare pass is looked up in the sorted anchor array (binary search,
// L2), a match finds its read in the offsets (binary search) and appends it once (a flag per read).
//
// Bound: HBM (1 byte per base read once; the look-behind of a thread is the line its neighbour streams).  VALU: ~25 operations per
// base, a quarter of what the sketch kernels spend.
#include "device_common.h"
#include "kernels.h"
#include <algorithm>

namespace drprg {
namespace dev {

namespace {

constexpr int AS_THREADS = 256;
constexpr int AS_SPAN = 32; // k-mer end positions per thread (and bases of look-behind)

__global__ __launch_bounds__(AS_THREADS) void anchor_scan_kernel(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
    uint32_t n_reads, uint64_t n_bases, const uint64_t* __restrict__ anchors, uint32_t n_anchors, uint32_t A, const uint32_t* __restrict__ prefilter,
    uint32_t batch, uint32_t* __restrict__ flags, unsigned long long* __restrict__ count, SelectedRead* __restrict__ list, uint64_t list_cap)
{
    __shared__ uint32_t pf[2048];
    for (int i = threadIdx.x; i < 2048; i += AS_THREADS) pf[i] = prefilter[i];
    __syncthreads();
    const uint64_t mask = A >= 32 ? ~0ull : ((1ull << (2 * A)) - 1);
    const uint64_t n_spans = (n_bases + AS_SPAN - 1) / AS_SPAN;
    for (uint64_t span = (uint64_t)blockIdx.x * AS_THREADS + threadIdx.x; span < n_spans; span += (uint64_t)gridDim.x * AS_THREADS) {
        const uint64_t s = span * AS_SPAN; // first end position of this thread
        uint32_t w[16];                    // bases s-32 .. s+31, four per word
        {
            const uint4* p = reinterpret_cast<const uint4*>(bases + s);
            uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0; // (zero bytes are no bases: the run restarts)
            if (s) {
                b0 = p[-2];
                b1 = p[-1];
            }
            const uint4 b2 = p[0], b3 = p[1]; // (the staging buffers end 64 bytes after the last base)
            w[0] = b0.x, w[1] = b0.y, w[2] = b0.z, w[3] = b0.w, w[4] = b1.x, w[5] = b1.y, w[6] = b1.z, w[7] = b1.w;
            w[8] = b2.x, w[9] = b2.y, w[10] = b2.z, w[11] = b2.w, w[12] = b3.x, w[13] = b3.y, w[14] = b3.z, w[15] = b3.w;
        }
        uint64_t v = 0;
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < 2 * AS_SPAN; ++i) {
            const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const uint32_t u = c & 0xDFu; // upper case
            const bool base = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
            const uint32_t code = ((c >> 1) & 3u) ^ ((c >> 2) & 1u); // A 0, C 1, G 2, T 3
            v = ((v << 2) | code) & mask;
            run = base ? run + 1 : 0;
            if (i < AS_SPAN) continue;
            const uint64_t end = s + (uint64_t)(i - AS_SPAN); // position of this base in the stream
            if (run < A || end >= n_bases) continue;
            const uint32_t low = (uint32_t)v & 0xFFFFu;
            if (!((pf[low >> 5] >> (low & 31)) & 1u)) continue;
            // sorted anchors: is v one of them?
            uint32_t lo = 0, hi = n_anchors;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (anchors[mid] < v) lo = mid + 1;
                else hi = mid;
            }
            if (lo >= n_anchors || anchors[lo] != v) continue;
            // the read the k-mer starts in: last r with offsets[r] <= start
            const uint64_t start = end + 1 - A;
            uint32_t a = 0, b = n_reads; // offsets[a] <= start < offsets[b]
            while (b - a > 1) {
                const uint32_t mid = (a + b) >> 1;
                if (offsets[mid] <= start) a = mid;
                else b = mid;
            }
            if (atomicExch(&flags[a], 1u) != 0u) continue;
            const unsigned long long at = atomicAdd(count, 1ull);
            if (at < list_cap) {
                SelectedRead sr;
                sr.offset = offsets[a];
                sr.len = (uint32_t)(offsets[a + 1] - offsets[a]);
                sr.read = a;
                sr.batch = batch;
                sr.pad = 0;
                list[at] = sr;
            }
        }
    }
}

// one wave per selected read: bytes from the batch it lives in to its place in the dense output
__global__ __launch_bounds__(256) void gather_reads_kernel(const GatherEntry* __restrict__ table, uint32_t n, uint8_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const GatherEntry e = table[i];
    for (uint32_t j = lane; j < e.len; j += 64) out[e.dst + j] = e.src[j];
}

} // namespace

hipError_t launch_anchor_scan(const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, uint64_t n_bases, const uint64_t* anchors,
    uint32_t n_anchors, uint32_t A, const uint32_t* prefilter, uint32_t batch, uint32_t* flags, unsigned long long* count, SelectedRead* list,
    uint64_t list_cap, int n_cus, hipStream_t stream)
{
    if (n_reads == 0 || n_bases == 0) return hipSuccess;
    const uint64_t n_spans = (n_bases + AS_SPAN - 1) / AS_SPAN;
    const uint64_t want = (n_spans + AS_THREADS - 1) / AS_THREADS;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)n_cus * 16);
    hipLaunchKernelGGL(anchor_scan_kernel, dim3(grid), dim3(AS_THREADS), 0, stream, bases, offsets, n_reads, n_bases, anchors, n_anchors, A, prefilter,
        batch, flags, count, list, list_cap);
    return hipGetLastError();
}

hipError_t launch_gather_reads(const GatherEntry* table, uint32_t n, uint8_t* out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_reads_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, table, n, out);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
