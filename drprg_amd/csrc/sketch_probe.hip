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
#include "device_common.h"

namespace drprg {
namespace dev {

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
    __shared__ uint32_t s_scan[SK_THREADS / 64 + 1];
    __shared__ unsigned long long s_base;
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
            uint4 in = load_once_16(a.bases + g);
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
                    HT hf, hr;
                    if constexpr (KC > 0 && sizeof(HT) == 4) {
                        hf = mix_k<KC>((uint32_t)fwd);
                        hr = mix_k<KC>((uint32_t)rev);
                    } else {
                        hf = Tr::mix(fwd, mask);
                        hr = Tr::mix(rev, mask);
                    }
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
    uint32_t minbits = 0;
    if (base0 >= halo && base0 < SK_NPOS - halo) {
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
    }
    { // the minimizers of the tile, compacted in position order (threads own consecutive positions)
        uint32_t n_all;
        uint32_t at = block_exclusive_scan<SK_THREADS / 64>((uint32_t)__popc(minbits), s_scan, &n_all);
        while (minbits) {
            int j = __ffs(minbits) - 1;
            minbits &= minbits - 1;
            s_mins[at++] = (uint16_t)(base0 + j);
        }
        if (tid == 0) s_nmin = n_all;
    }
    __syncthreads();

    // ---- phase 2b: probe the index with the compacted minimizer list ----
    // Found slots replace the hash in s_hash (same thread, same word); the hits of the whole tile then take ONE global
    // reservation (a per-minimizer atomic on the hit counter serialises on a single L2 address once millions of
    // minimizers hit, as they do with a large index).
    const uint32_t nmin = s_nmin;
    const double reads_per_base = (double)a.n_reads / (double)(a.n_bases ? a.n_bases : 1);
    const uint32_t w1_magic = w1_reciprocal(w);
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const HT* __restrict__ slot_key = reinterpret_cast<const HT*>(a.slot_key);
    constexpr HT NOT_FOUND = (HT)~(HT)0;
    if (a.tile_cap) {
        // ---- candidate form: one record per index minimizer, in position order, into this tile's slice ----
        const size_t slice = (size_t)blockIdx.x * a.tile_cap;
        uint32_t written = 0, my_hits = 0;
        for (uint32_t i0 = 0; i0 < nmin; i0 += SK_THREADS) { // (wave-uniform trip count: the scans hold barriers)
            const uint32_t i = i0 + (uint32_t)tid;
            bool found = false;
            uint32_t s = 0, guess = 0;
            int j = 0;
            uint64_t gp = 0, o0 = 0, o1 = 0;
            uint4 sf = make_uint4(0, 0, 0, 0);
            if (i < nmin) {
                j = s_mins[i];
                const HT h = s_hash[hpad(j)] - 1;
                // the read lookup does not depend on the probe: its two loads go out first and overlap the probe's chain
                gp = (uint64_t)(origin + j);
                guess = (uint32_t)((double)gp * reads_per_base);
                if (guess >= a.n_reads) guess = a.n_reads - 1;
                o0 = a.offsets[guess];
                o1 = a.offsets[guess + 1];
                bool maybe = true;
                if (a.pbloom) {
                    const uint32_t m = pbloom_mix((uint64_t)h), need = pbloom_bits(m);
                    maybe = (a.pbloom[pbloom_word(m, a.pbloom_wbits)] & need) == need;
                }
                if (maybe) {
                    s = table_slot_dev(h, a.table_bits);
                    while (true) {
                        const HT key = slot_key[s];
                        if (key == h) { found = true; break; }
                        if (key == Tr::EMPTY) break;
                        s = (s + 1) & tmask;
                    }
                }
                if (found) sf = a.slot_first[s]; // (in flight across the scan's barriers)
            }
            uint32_t n_found;
            const uint32_t at = written + block_exclusive_scan<SK_THREADS / 64>(found ? 1u : 0u, s_scan, &n_found);
            if (found) {
                const uint2 rec = make_uint2(sf.x, sf.y); // record offset, count; sf.z the first record's node, sf.w its prg | shortest path << 12
                uint32_t read = guess;
                if (!(o0 <= gp && gp < o1)) { // interpolation missed (reads of different lengths): gallop from the tile's first read
                    read = find_read_from(a.offsets, a.n_reads, first_read ? first_read - 1 : 0, gp);
                    o0 = a.offsets[read];
                    o1 = a.offsets[read + 1];
                }
                const uint64_t r0 = o0, r1 = o1, pos = gp - r0;
                const uint32_t strand = (s_strand[j / SK_G] >> (j % SK_G)) & 1u;
                my_hits += rec.y;
                if (pos >= (1ull << HIT_POS_BITS)) atomicOr(a.overflow, 2u);
                else if (at < a.tile_cap) {
                    const uint32_t kn = sf.z, prg = sf.w & 0xFFFu;
                    const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                    // size threshold of a cluster of this read on that PRG (cluster_eval_kernel)
                    const uint64_t expected = expected_minimizers(r1 - r0, w, w1_magic);
                    uint64_t m = sf.w >> 12;
                    if (expected < m) m = expected;
                    const uint32_t length_based = (uint32_t)((double)m * a.fraction);
                    uint32_t thr = length_based > a.min_cluster_size ? length_based : a.min_cluster_size;
                    if (thr > 0xFFFFu) thr = 0xFFFFu;
                    a.tile_info[slice + at] = ((uint64_t)s << 32) | ((uint64_t)strand << 31) | (uint64_t)read;
                    a.tile_pos1[slice + at] = (uint32_t)pos + 1;
                    a.tile_rec[slice + at] = make_uint4(rec.x, rec.y, (strand << 31) | (((prg << 1) | rev) << 16) | thr, (kn >> 1) * 2u + rev);
                }
            }
            written += n_found;
        }
        uint32_t tile_hits;
        (void)block_exclusive_scan<SK_THREADS / 64>(my_hits, s_scan, &tile_hits);
        if (tid == 0) {
            a.tile_count[blockIdx.x] = written < a.tile_cap ? written : a.tile_cap;
            a.tile_hits[blockIdx.x] = tile_hits;
            a.tile_nmin[blockIdx.x] = nmin; // (summed by tile_totals_kernel: one atomic per tile on one counter would be ~15 ns each, in series)
            if (written > a.tile_cap) atomicOr(a.overflow, 4u);
        }
        return;
    }
    uint32_t mine = 0;
    for (uint32_t i = tid; i < nmin; i += SK_THREADS) {
        const int j = s_mins[i];
        const HT h = s_hash[hpad(j)] - 1;
        if (a.pbloom) { // three of four minimizers are no index keys: they stop at one L2-resident word
            const uint32_t m = pbloom_mix((uint64_t)h), need = pbloom_bits(m);
            if ((a.pbloom[pbloom_word(m, a.pbloom_wbits)] & need) != need) {
                s_hash[hpad(j)] = NOT_FOUND;
                continue;
            }
        }
        uint32_t s = table_slot_dev(h, a.table_bits);
        bool found = false;
        while (true) {
            const HT key = slot_key[s];
            if (key == h) { found = true; break; }
            if (key == Tr::EMPTY) break;
            s = (s + 1) & tmask;
        }
        s_hash[hpad(j)] = found ? (HT)s : NOT_FOUND;
        if (found) mine += a.slot_rec[s].y;
    }
    uint32_t total;
    const uint32_t before = block_exclusive_scan<SK_THREADS / 64>(mine, s_scan, &total);
    if (tid == 0) s_base = total ? atomicAdd(a.n_hits, (unsigned long long)total) : 0ull;
    __syncthreads();
    unsigned long long at = s_base + before;
    for (uint32_t i = tid; i < nmin; i += SK_THREADS) {
        const int j = s_mins[i];
        const HT sv = s_hash[hpad(j)];
        if (sv == NOT_FOUND) continue;
        const uint32_t s = (uint32_t)sv;
        const uint2 rec = a.slot_rec[s];
        // a hit: locate the read and emit one hit per index record
        const uint64_t gp = (uint64_t)(origin + j);
        const uint32_t read = find_read_guess(a.offsets, a.n_reads, (uint32_t)((double)gp * reads_per_base), first_read ? first_read - 1 : 0, gp);
        const uint64_t pos = gp - a.offsets[read];
        const uint32_t strand = (s_strand[j / SK_G] >> (j % SK_G)) & 1u;
        const unsigned long long mine_at = at;
        at += rec.y;
        if (mine_at + rec.y > a.hit_capacity || pos >= (1ull << HIT_POS_BITS)) {
            atomicOr(a.overflow, pos >= (1ull << HIT_POS_BITS) ? 2u : 1u);
            continue;
        }
        for (uint32_t q = 0; q < rec.y; ++q) {
            const uint32_t kn = a.rec_knode[rec.x + q]; // (global knode << 1) | strand
            const uint32_t prg = a.rec_prg[rec.x + q];
            const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u; // forward hits sort first
            a.hit_key[mine_at + q] = pack_hit_key(read, prg, rev, (uint32_t)pos);
            a.hit_val[mine_at + q] = kn >> 1;
        }
    }
    if (tid == 0 && nmin) atomicAdd(a.n_minimizers, (unsigned long long)nmin);
}

hipError_t launch_tile_first_read(const uint64_t* offsets, uint32_t n_reads, int t_eval, int halo, uint32_t n_tiles, uint32_t* out,
    hipStream_t stream)
{
    hipLaunchKernelGGL(tile_first_read_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, offsets, n_reads, t_eval, halo, n_tiles, out);
    return hipGetLastError();
}

uint32_t sketch_tile_eval(int halo) { return (uint32_t)(SK_NPOS - 2 * halo); }

uint32_t sketch_n_tiles(uint64_t n_bases, int halo)
{
    uint64_t t_eval = sketch_tile_eval(halo);
    return (uint32_t)((n_bases + t_eval - 1) / t_eval);
}

hipError_t launch_sketch_probe(const SketchArgs& a, bool wide_hash, hipStream_t stream, KernelTimer timer)
{
    if (a.n_bases == 0) return hipSuccess;
    const uint32_t grid = sketch_n_tiles(a.n_bases, a.halo); // positions past n_bases-k are invalid inside the kernel
    HIP_TRY(launch_tile_first_read(a.offsets, a.n_reads, (int)sketch_tile_eval(a.halo), a.halo, grid, a.tile_first_read, stream));
    const dim3 g(grid), b(SK_THREADS);
    if (wide_hash)
        launch_timed(timer, sketch_probe_kernel<uint64_t, 0, 0>, g, b, 0, stream, a);
    else if (a.k == 15 && a.w == 11)
        launch_timed(timer, sketch_probe_kernel<uint32_t, 15, 11>, g, b, 0, stream, a);
    else if (a.k == 15 && a.w == 14)
        launch_timed(timer, sketch_probe_kernel<uint32_t, 15, 14>, g, b, 0, stream, a);
    else
        launch_timed(timer, sketch_probe_kernel<uint32_t, 0, 0>, g, b, 0, stream, a);
    HIP_TRY(hipGetLastError());
    return hipSuccess;
}

} // namespace dev
} // namespace drprg
