// candidates.hip -- the candidate stage of the filtered launch sequence (see the overview at the top of
// sketch_filter.hip): slices of candidate positions -> one dense list ordered by (read, position) -> exact lookup and
// window-minimizer test per candidate (verify_count_kernel) -> batch totals; and, for the reads that
// read_cluster_kernel leaves over, the hit list for the generic cluster pipeline (recount / expand / per-read reorder).
#include "filter_common.h"
#include "verify_lane.h"
#include <algorithm>
#include <cstdlib>
#include <string>
#include <cstdint>

namespace drprg {
namespace dev {

// ---------------------------------------------------------------------------------------------
// scans
// ---------------------------------------------------------------------------------------------
// one workgroup: cand_prefix = exclusive scan of slice_count (the gathered form: DRPRG_VERIFY_FORM=gather, read_verify_kernel)
__global__ __launch_bounds__(SCAN_THREADS) void cand_scan_kernel(FilterWork fw)
{
    __shared__ uint32_t s_w[SCAN_THREADS / 64 + 1];
    constexpr int PER = MAX_CHUNKS / SCAN_THREADS;
    const int tid = threadIdx.x;
    // two passes over this thread's PER consecutive slices: their sum, then -- behind the block scan -- their prefixes
    uint32_t run = 0;
    for (int i = 0; i < PER; ++i) {
        const uint32_t s = (uint32_t)tid * PER + i;
        if (s >= fw.n_slices) break;
        const uint32_t n = fw.slice_count[s];
        run += n;
    }
    uint32_t total;
    uint32_t acc = block_exclusive_scan<SCAN_THREADS / 64>(run, s_w, &total);
    for (int i = 0; i < PER; ++i) {
        const uint32_t s = (uint32_t)tid * PER + i;
        if (s >= fw.n_slices) break;
        const uint32_t n = fw.slice_count[s];
        fw.cand_prefix[s] = acc;
        acc += n;
    }
    if (tid == 0) fw.cand_prefix[fw.n_slices] = total;
}

// ---------------------------------------------------------------------------------------------
// verification (the per-candidate device functions: verify_lane.h)
// ---------------------------------------------------------------------------------------------
// slices -> one dense, ordered list of candidate positions (cand_gp; the verification kernel writes cand_info / cand_pos1 / cand_rec
// at the same indices.  Until round 5 the positions went to cand_info and were replaced there: read_verify_kernel's workgroups read
// positions their neighbours own, so the list has to stay as it is)
__global__ __launch_bounds__(64) void cand_gather_kernel(FilterWork fw)
{
    const uint32_t s = blockIdx.x;
    const uint32_t n = fw.cand_prefix[s + 1] - fw.cand_prefix[s];
    const uint64_t* __restrict__ src = fw.raw_pos + fw.slice_base[s];
    uint64_t* __restrict__ dst = fw.cand_gp + fw.cand_prefix[s];
    for (uint32_t i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}

template <int KC, bool PACKED>
__global__ __launch_bounds__(EX_THREADS) void verify_count_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    __shared__ uint32_t s_red[3][EX_THREADS / 64];
    const int tid = threadIdx.x;
    uint32_t t_begin, t_end;
    candidate_range(fw, blockIdx.x, gridDim.x, t_begin, t_end);
    const VerifyConsts c(a, fw);
    uint32_t my_hits = 0, my_nmin = 0, my_maxlen = 0;
    // (the position of this thread's next candidate is requested one round early: one round trip less in the chain of each)
    int64_t gp_next = t_begin + tid < t_end ? (int64_t)fw.cand_gp[t_begin + tid] : 0;
    for (uint32_t t = t_begin + tid; t < t_end; t += EX_THREADS) {
        const int64_t gp = gp_next;
        if (t + EX_THREADS < t_end) gp_next = (int64_t)fw.cand_gp[t + EX_THREADS];
        VerifyOut o;
        verify_one_lane<KC, PACKED>(a, fw, rc, c, gp, o, my_hits, my_nmin, my_maxlen);
        fw.cand_pos1[t] = o.pos1;
        fw.cand_info[t] = ((uint64_t)o.slot << 32) | ((uint64_t)o.strand << 31) | (uint64_t)o.read;
        fw.cand_rec[t] = o.crec;
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

// The same kernel WITHOUT cand_scan_kernel and cand_gather_kernel in front of it (round 5: 10 + 12 us of a 0.59 ms step, one workgroup's
// latency and a copy of 14 MB).  Every workgroup scans the counts itself -- since round 6 those of the SUPERBLOCKS, FT_SUPER slices each, which
// the filter kernel keeps next to the slice counts: 8192 words at most, 16 per thread, one coalesced round trip and a block scan: 16 MB of L2
// reads over the whole grid --, leaves the exclusive prefix of every superblock in LDS (32 KB), takes its share [t_begin, t_end) of the
// ordered list from the total, and reads the positions of its candidates straight from the filter kernel's slices: entry t lives in the
// superblock s with prefix[s] <= t < prefix[s + 1] (a candidate bisects the superblocks of its workgroup's share, a handful of a full batch),
// there in the slice the eight slice counts of the superblock say (one more round trip to the L2, requested a candidate ahead like the
// position itself), at raw_pos[start of the slice + rest].  Workgroup 0 leaves the total where read_cluster_kernel and the generic pipeline
// look for it (*fw.cand_total).  The dense list of positions (fw.cand_gp) is not made: only the experimental read-by-read form wants it,
// and gets the old sequence.
#ifndef DRPRG_VS_THREADS // (measurement builds)
#define DRPRG_VS_THREADS 512
#endif
// threads per workgroup: every workgroup pays the slice scan once, so fewer and larger ones pay less of it -- 256 (as verify_count_kernel, 8 per
// CU) measured 0.516 ms per step on the 8d index (packed 0.452), 512 (4 per CU) 0.503-0.512 (0.445), 1024 (2 per CU) 0.507-0.514 (0.447);
// the larger indexes do not care (profiles/r05/verify_scan.txt)
constexpr int VS_THREADS = DRPRG_VS_THREADS;
constexpr int VS_PER = MAX_SLICES / VS_THREADS; // superblocks per thread of the scan
template <int KC, bool PACKED>
__global__ __launch_bounds__(VS_THREADS, 8) void verify_scan_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc) // (8 waves per SIMD = 64 VGPRs: the ASCII form wants 74 and spills 28 bytes per lane, and is still faster for the eighth wave: 102 -> 97 us, nanopore 424 -> 383 us)
{
    __shared__ uint32_t s_red[3][VS_THREADS / 64];
    __shared__ uint32_t s_w[VS_THREADS / 64 + 1];
    __shared__ uint32_t s_share[2];
    __shared__ uint32_t s_ticket; // (interleaved order: the workgroup's round counter)
    __shared__ __attribute__((aligned(16))) uint32_t s_pre[MAX_SLICES + 4]; // exclusive prefix of ALL superblocks (32 KB; one past the last: the total)
    const int tid = threadIdx.x;
    // ---- the scan: my VS_PER consecutive superblocks (FT_SUPER slices each; the filter kernel summed their clamped counts) ----
    static_assert(VS_PER % 4 == 0 && MAX_SLICES % (VS_THREADS * 4) == 0, "the counts are read 16 bytes at a time");
    auto my_count = [&](int i4) -> uint4 { // counts 4 i4 .. 4 i4 + 3 of this thread's superblocks (those past the last slice are zero)
        return reinterpret_cast<const uint4*>(fw.super_count + (size_t)tid * VS_PER)[i4]; // (16-byte aligned: filter_small_words)
    };
    uint32_t run = 0;
    uint4 v[VS_PER / 4];
#pragma unroll
    for (int i = 0; i < VS_PER / 4; ++i) v[i] = my_count(i); // (all loads before the first sum)
#pragma unroll
    for (int i = 0; i < VS_PER / 4; ++i) run += v[i].x + v[i].y + v[i].z + v[i].w;
    uint32_t total;
    const uint32_t before = block_exclusive_scan<VS_THREADS / 64>(run, s_w, &total);
    if (blockIdx.x == 0 && tid == 0) fw.cand_prefix[fw.n_slices] = total; // = *fw.cand_total
    if (fw.debug & 2048u) return; // (DRPRG_FT_DEBUG=2048: measurement only, the slice scan alone)
    // Every thread leaves the prefixes of its own slices in LDS, out of the counts it still holds: entry t of the list then lives in the
    // slice s with s_pre[s] <= t < s_pre[s + 1], for every t, and no workgroup reads a count twice.  (Until late in round 5 a window of 64
    // prefixes was rebuilt -- one more dependent load, a wave scan and two barriers: 3-4.5 us per workgroup -- every 64 slices of its share.)
    const uint32_t per_wg = (total + gridDim.x - 1) / gridDim.x;
    const uint64_t b64 = (uint64_t)blockIdx.x * per_wg;
    const uint32_t t_begin = b64 < total ? (uint32_t)b64 : total, t_end = b64 + per_wg < total ? (uint32_t)(b64 + per_wg) : total;
    {
        // (... and the two slices that bound this workgroup's share come from the threads that hold them: the number of a thread's prefixes
        // that do not exceed the entry, counted in registers -- instead of two 13-step searches of dependent LDS reads in every lane)
        const uint32_t x0 = t_begin, x1 = t_end - 1u; // (x1 only used when t_begin < t_end)
        const bool own0 = t_begin < t_end && before <= x0 && x0 < before + run, own1 = t_begin < t_end && before <= x1 && x1 < before + run;
        uint32_t acc = before, n0 = 0, n1 = 0;
        uint4* dst = reinterpret_cast<uint4*>(s_pre + (size_t)tid * VS_PER);
#pragma unroll
        for (int i = 0; i < VS_PER / 4; ++i) {
            uint4 p;
            p.x = acc;
            p.y = p.x + v[i].x;
            p.z = p.y + v[i].y;
            p.w = p.z + v[i].z;
            acc = p.w + v[i].w;
            dst[i] = p;
            // prefixes of my slices 4 i + 1 .. 4 i + 4 (the last one: the next thread's first, > any entry I own)
            n0 += (p.y <= x0 ? 1u : 0u) + (p.z <= x0 ? 1u : 0u) + (p.w <= x0 ? 1u : 0u) + (acc <= x0 ? 1u : 0u);
            n1 += (p.y <= x1 ? 1u : 0u) + (p.z <= x1 ? 1u : 0u) + (p.w <= x1 ? 1u : 0u) + (acc <= x1 ? 1u : 0u);
        }
        if (tid == VS_THREADS - 1) s_pre[MAX_SLICES] = acc; // (= total)
        if (tid == 0) s_ticket = 0;
        if (own0) s_share[0] = (uint32_t)tid * VS_PER + n0;
        if (own1) s_share[1] = (uint32_t)tid * VS_PER + n1;
    }
    __syncthreads();
    const VerifyConsts c(a, fw);
    uint32_t my_hits = 0, my_nmin = 0, my_maxlen = 0;
    // entry t of the ordered list, given superblocks lo0 <= .. <= hi0 that bound it and the largest power of two <= hi0 - lo0 (at least 1)
    auto position_in = [&](uint32_t t, uint32_t lo0, uint32_t hi0, uint32_t step0) -> int64_t {
        uint32_t lo = lo0;
        for (uint32_t step = step0; step >= 1; step >>= 1)
            if (lo + step <= hi0 && s_pre[lo + step] <= t) lo += step;
        // ... in superblock lo; its slice: the counts and the starts of the FT_SUPER slices (four 16-byte loads from the L2, one round trip), then
        // the first one whose running sum exceeds the rest.  (Counts past the last slice of the batch are never reached: the superblock sums say where the list ends.)
        static_assert(FT_SUPER == 8, "two uint4 of slice counts per superblock");
        const uint4* __restrict__ c4 = reinterpret_cast<const uint4*>(fw.slice_count) + (size_t)lo * 2;
        const uint4* __restrict__ b4 = reinterpret_cast<const uint4*>(fw.slice_base) + (size_t)lo * 2;
        const uint4 ca = c4[0], cb = c4[1], ba = b4[0], bb = b4[1];
        const uint32_t cnt[7] = { ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z }, start[8] = { ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w };
        uint32_t rest = t - s_pre[lo], sum = 0, before = 0, at = start[0];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            sum += cnt[q];
            const bool past = rest >= sum;
            at = past ? start[q + 1] : at;
            before = past ? sum : before;
        }
        return (int64_t)fw.raw_pos[(size_t)at + (rest - before)];
    };
    if (fw.verify_interleave) {
        // Interleaved order (round 6): the list is cut into rounds of 64 consecutive candidates; round r belongs to workgroup r mod grid, whose
        // waves take its rounds by ticket (an LDS counter; a wave's first is its own number).  A workgroup samples the whole list instead
        // of owning one stretch of it -- stretches differ: reads off the panel end at the table probe, reads on it walk their windows --
        // and a wave that drew cheap rounds draws more of them.
        const uint32_t n_rounds = (total + 63u) >> 6;
        const uint32_t lane = (uint32_t)tid & 63u;
        uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(tid >> 6) * gridDim.x + blockIdx.x));
        int64_t gp_next = (r < n_rounds && r * 64u + lane < total) ? position_in(r * 64u + lane, 0u, (uint32_t)MAX_SLICES - 1u, (uint32_t)MAX_SLICES / 2u) : 0;
        while (r < n_rounds) { // (wave-uniform)
            const uint32_t t = r * 64u + lane;
            const int64_t gp = gp_next;
            uint32_t drawn = 0;
            if (lane == 0) drawn = atomicAdd(&s_ticket, 1u);
            const uint32_t rn = ((uint32_t)(VS_THREADS / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn)) * gridDim.x + blockIdx.x;
            if (rn < n_rounds && rn * 64u + lane < total) gp_next = position_in(rn * 64u + lane, 0u, (uint32_t)MAX_SLICES - 1u, (uint32_t)MAX_SLICES / 2u);
            if (t < total) {
                VerifyOut o;
                if (!(fw.debug & 512u)) verify_one_lane<KC, PACKED>(a, fw, rc, c, gp, o, my_hits, my_nmin, my_maxlen);
                fw.cand_pos1[t] = o.pos1;
                fw.cand_info[t] = ((uint64_t)o.slot << 32) | ((uint64_t)o.strand << 31) | (uint64_t)o.read;
                fw.cand_rec[t] = o.crec;
            }
            r = rn;
        }
    } else if (t_begin < t_end) { // (workgroup-uniform)
        // the superblocks of this workgroup's share [t_begin, t_end): its candidates bisect only those (a handful of a full batch)
        const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_share[0]), s_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_share[1]);
        uint32_t span_step = 1;
        while (2 * span_step <= s_hi - s_lo) span_step *= 2; // (largest power of two <= the span, at least 1)
        auto position_of = [&](uint32_t t) -> int64_t { return position_in(t, s_lo, s_hi, span_step); }; // t_begin <= t < t_end
        // (the position of this thread's next candidate is requested one round early: one round trip less in the chain of each)
        int64_t gp_next = t_begin + tid < t_end ? position_of(t_begin + (uint32_t)tid) : 0;
        for (uint32_t t = t_begin + (uint32_t)tid; t < t_end; t += VS_THREADS) {
            const int64_t gp = gp_next;
            if (t + VS_THREADS < t_end) gp_next = position_of(t + VS_THREADS);
            VerifyOut o;
            if (!(fw.debug & 512u)) verify_one_lane<KC, PACKED>(a, fw, rc, c, gp, o, my_hits, my_nmin, my_maxlen); // (DRPRG_FT_DEBUG=512: measurement only, the scan and the positions alone)
            fw.cand_pos1[t] = o.pos1;
            fw.cand_info[t] = ((uint64_t)o.slot << 32) | ((uint64_t)o.strand << 31) | (uint64_t)o.read;
            fw.cand_rec[t] = o.crec;
        }
    }
    // ---- per-workgroup totals ----
    const uint32_t wh = wave_inclusive_scan(my_hits), wn = wave_inclusive_scan(my_nmin), wm = wave_max(my_maxlen);
    if ((tid & 63) == 63) {
        s_red[0][tid >> 6] = wh;
        s_red[1][tid >> 6] = wn;
        s_red[2][tid >> 6] = wm;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t h = 0, n = 0, mx = 0;
        for (int i = 0; i < VS_THREADS / 64; ++i) {
            h += s_red[0][i];
            n += s_red[1][i];
            mx = s_red[2][i] > mx ? s_red[2][i] : mx;
        }
        fw.wg_hits[blockIdx.x] = h;
        fw.wg_nmin[blockIdx.x] = n;
        fw.wg_maxlen[blockIdx.x] = mx;
    }
}

// (Round 3 built a wave-cooperative form of this kernel -- the 64 consecutive candidates of a wave work out which k-mer positions
// their windows need, hash each ONCE, four per lane, into LDS, and read their neighbours from there -- and measured it slower than
// one lane per candidate on every index size: 151 against 118-132 us on the 8d index, 1.06 against 0.82 ms on the 8-fold one.  The
// SQ counters say why: 51.6 M VALU wave-instructions against 54.3 M.  Stray candidates and read ends keep the shared windows short
// (19 hashes per candidate instead of 42), hashing is under half of the lane form's instructions, and the bookkeeping -- segments,
// quad origins, 64-bit positions, LDS round trips -- costs what the sharing saves.  DESIGN.md section 6.)

// (... and a nearest-first form: every wave takes 128 candidates, two per lane, scans the w-1 LEFT-hand neighbours of each, emits those a
// window ending with them already settles -- 54 % of the true minimizers --, parks the others in a queue of its own in LDS and finishes
// them 64 at a time with w-1 RIGHT-hand steps.  Fewer hashes (~31 instead of 42 steps per 2 candidates) and bit-identical results, but
// 71 VGPRs instead of 54, 18 KB of LDS per workgroup and the queue traffic: 115.7 us against 114-116 on the 8d index, 895 against 825 us
// on the 8-fold one.  Not kept.)

// one workgroup: wg_base = exclusive scan of wg_hits; batch totals
// n_wg: the workgroups whose totals wg_hits / wg_nmin / wg_maxlen hold (the verification kernel's grid, or recount_kernel's)
__global__ __launch_bounds__(SCAN_THREADS) void hit_scan_kernel(SketchArgs a, FilterWork fw, int recount, uint32_t n_wg)
{
    __shared__ uint32_t s_w[SCAN_THREADS / 64 + 1];
    __shared__ uint32_t s_n[SCAN_THREADS / 64], s_m[SCAN_THREADS / 64];
    constexpr int PER = MAX_EX_WG / SCAN_THREADS;
    const int tid = threadIdx.x;
    uint32_t v[PER], h[PER], nm[PER], ml[PER], run = 0, nmin = 0, mx = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { // (all twelve loads together, as in cand_scan_kernel)
        const uint32_t g = (uint32_t)tid * PER + i, gc = g < n_wg ? g : 0u;
        h[i] = fw.wg_hits[gc];
        nm[i] = fw.wg_nmin[gc];
        ml[i] = fw.wg_maxlen[gc];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t g = (uint32_t)tid * PER + i;
        v[i] = run;
        if (g < n_wg) {
            run += h[i];
            nmin += nm[i];
            mx = ml[i] > mx ? ml[i] : mx;
        }
    }
    uint32_t total;
    const uint32_t before = block_exclusive_scan<SCAN_THREADS / 64>(run, s_w, &total);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t g = (uint32_t)tid * PER + i;
        if (g < n_wg) fw.wg_base[g] = before + v[i];
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
        // (a sequence whose candidate slices overflowed is run again by the host: it must not count twice)
        if (n && !recount && !(*reinterpret_cast<volatile uint32_t*>(a.overflow) & 4u)) atomicAdd(a.n_minimizers, (unsigned long long)n);
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

// ---------------------------------------------------------------------------------------------
// candidate form of the direct sequence: tile slices -> dense list
// ---------------------------------------------------------------------------------------------
constexpr int TG_THREADS = 1024, TG_TILES = 64; // tiles per workgroup: every wave copies four tile slices
// (a slice or the dense list too small: the host grows the workspace and runs the batch again; nothing is counted or copied)
__device__ __forceinline__ bool tile_overflow(const SketchArgs& a, const uint32_t* tile_prefix, uint32_t n_tiles, uint64_t dense_capacity)
{
    return (*reinterpret_cast<volatile uint32_t*>(a.overflow) & 4u) != 0 || (uint64_t)tile_prefix[n_tiles] > dense_capacity;
}

// The batch counters: hits and minimizers of all tiles (and what sketch_wave_kernel clustered itself), summed by a small grid --
// one atomic per counter and workgroup: as part of the gather (6000 workgroups, four atomics each on the same four addresses) the
// sums cost 0.3 ms of its 0.8 on the 500-locus index.
// block_first != nullptr (read_cluster_kernel<SLICES> follows): word m = the slice that holds entry 64 m of the ordered list
__global__ __launch_bounds__(TG_THREADS) void tile_totals_kernel(SketchArgs a, const uint32_t* __restrict__ tile_prefix, uint32_t n_tiles,
    uint64_t dense_capacity, uint32_t* __restrict__ block_first)
{
    __shared__ uint32_t s_w[TG_THREADS / 64 + 1];
    const int tid = threadIdx.x;
    if (tile_overflow(a, tile_prefix, n_tiles, dense_capacity)) {
        if (blockIdx.x == 0 && tid == 0) atomicOr(a.overflow, 4u);
        return;
    }
    uint32_t my_hits = 0, my_nmin = 0, my_fc = 0, my_fh = 0;
    for (uint32_t t = blockIdx.x * TG_THREADS + (uint32_t)tid; t < n_tiles; t += gridDim.x * TG_THREADS) {
        my_hits += a.tile_hits[t];
        my_nmin += a.tile_nmin[t];
        if (block_first) {
            const uint32_t lo = tile_prefix[t], hi = tile_prefix[t + 1];
            for (uint32_t m = (lo + 63u) >> 6; lo < hi && (uint64_t)m << 6 < hi; ++m) block_first[m] = t;
        }
        if (a.fuse > 0) { // what sketch_wave_kernel clustered itself
            const uint32_t f = a.tile_fast[t];
            my_fc += f & 0xFFFFu;
            my_fh += f >> 16;
        }
    }
    // (sums of 32-bit partials per workgroup: a workgroup covers at most n_tiles / gridDim.x + 1024 tiles of <= 2^16 hits each)
    uint32_t hits, nmin, fc, fh;
    (void)block_exclusive_scan<TG_THREADS / 64>(my_hits, s_w, &hits);
    (void)block_exclusive_scan<TG_THREADS / 64>(my_nmin, s_w, &nmin);
    (void)block_exclusive_scan<TG_THREADS / 64>(my_fc, s_w, &fc);
    (void)block_exclusive_scan<TG_THREADS / 64>(my_fh, s_w, &fh);
    if (tid == 0) {
        if (hits) atomicAdd(a.n_hits, (unsigned long long)hits);
        if (nmin) atomicAdd(a.n_minimizers, (unsigned long long)nmin);
        if (fc) atomicAdd(a.n_clusters_kept, (unsigned long long)fc);
        if (fh) atomicAdd(a.n_hits_kept, (unsigned long long)fh);
    }
}

// The slices copied into the dense, ordered candidate list.  mark != 0 (a copy after read_cluster_kernel has read the candidates
// from the slices): a candidate whose dense cand_pos1 holds the mark was handled there and gets position 0.
__global__ __launch_bounds__(TG_THREADS) void tile_gather_kernel(SketchArgs a, FilterWork fw, const uint32_t* __restrict__ tile_prefix,
    uint32_t n_tiles, uint64_t dense_capacity, uint32_t mark)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tile_overflow(a, tile_prefix, n_tiles, dense_capacity)) return;
    const uint32_t t0 = blockIdx.x * TG_TILES, t1 = t0 + TG_TILES < n_tiles ? t0 + TG_TILES : n_tiles;
    for (uint32_t t = t0 + (uint32_t)wave; t < t1; t += TG_THREADS / 64) {
        const uint32_t n = a.tile_count[t], dst = tile_prefix[t];
        const size_t src = (size_t)t * a.tile_cap;
        // (256 records per round, all twelve loads before the first store -- a lane past the end reads record 0 and drops it --: a
        // slice holds about 150 records, and one load-store round trip per 64 of them left the memory system half idle)
        for (uint32_t i0 = 0; i0 < n; i0 += 256) {
            uint64_t ci[4];
            uint32_t cp[4];
            uint4 cr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
                const size_t at = src + (i < n ? i : 0u);
                ci[u] = a.tile_info[at];
                cp[u] = a.tile_pos1[at];
                cr[u] = a.tile_rec[at];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
                if (i < n) {
                    fw.cand_info[dst + i] = ci[u];
                    fw.cand_pos1[dst + i] = (mark && fw.cand_pos1[dst + i] == mark) ? 0u : cp[u];
                    fw.cand_rec[dst + i] = cr[u];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// (Round 3 tried to do without the two single-workgroup scans of this sequence, cand_scan_kernel and hit_scan_kernel, 10 + 6 us:
// every workgroup of cand_gather_kernel summing the slice counts before its own, and every workgroup of verify_count_kernel adding
// its totals to the batch counters with three atomics.  Measured: cand_gather 11 -> 38 us (8192 workgroups x up to 8192 loads),
// verify_count 116 -> 231 us (12 k atomics on one 64-byte line take their turn in the L2), step 0.59 -> 0.72 ms.  Round 4: the totals of
// hit_scan_kernel are summed by workgroup 0 of read_cluster_kernel when that kernel follows (with_totals = false); scan + gather in one
// launch -- a workgroup summing what lies before its 64 slices -- took 19.4 us against 9.1 + 11.8: cand_scan_kernel stays.)
// DRPRG_VERIFY_FORM=gather keeps the three-kernel sequence of rounds 1-4 (A/B runs, a second way through the parity tests); the
// experimental read-by-read form (make EXPERIMENTAL=1 + DRPRG_VERIFY_FORM=read: read_verify.hip, bit-exact and 8 % slower,
// profiles/r05/read_verify.txt) needs the gathered list of positions as well.  Read at every launch (the tests switch it); the host
// allocates FilterBuffers::cand_gp only for a launch that will gather (8 bytes per candidate slot the default sequence never touches: ADVICE r05).
bool gathered_list_requested()
{
    const char* form = std::getenv("DRPRG_VERIFY_FORM");
    if (!form) return false;
    const std::string f(form);
#ifdef DRPRG_EXPERIMENTAL
    return f == "gather" || f == "read";
#else
    return f == "gather";
#endif
}

hipError_t launch_candidate_stage(const SketchArgs& a, FilterWork& fw, const ReadClusterArgs& rc, int n_cus, hipStream_t stream, bool with_totals)
{
    fw.verify_grid = fw.ex_grid;
    const char* form = std::getenv("DRPRG_VERIFY_FORM");
    bool gathered = form && std::string(form) == "gather";
    {   // DRPRG_VERIFY_ORDER=contiguous | interleave (read at every launch): which candidates a workgroup of verify_scan_kernel takes
        const char* order = std::getenv("DRPRG_VERIFY_ORDER");
        fw.verify_interleave = order && std::string(order) == "interleave" ? 1u : 0u;
    }
#ifdef DRPRG_EXPERIMENTAL
    const bool by_read = read_verify_applies(a, fw);
    gathered = gathered || by_read;
#endif
    if (gathered && !fw.cand_gp) return hipErrorInvalidValue; // (the caller allocates it when gathered_list_requested())
    if (!gathered) { // one launch: verify_scan_kernel scans the slice counts itself and reads the slices
        // (DRPRG_VERIFY_WG_PER_CU: measurements.  The grid is what fills the CUs' 2048 thread slots: fewer or more workgroups per CU measured
        // within the noise or worse; profiles/r05/verify_scan.txt)
        static const int per_cu = [] { const char* e = std::getenv("DRPRG_VERIFY_WG_PER_CU"); return e ? std::max(1, std::atoi(e)) : 0; }();
        fw.verify_grid = std::min<uint32_t>((uint32_t)n_cus * (uint32_t)(per_cu ? per_cu : 2048 / VS_THREADS), MAX_EX_WG); // (2048 threads per CU)
        const dim3 grid(fw.verify_grid);
        if (a.packed) {
            if (a.k == 15) hipLaunchKernelGGL((verify_scan_kernel<15, true>), grid, dim3(VS_THREADS), 0, stream, a, fw, rc);
            else hipLaunchKernelGGL((verify_scan_kernel<0, true>), grid, dim3(VS_THREADS), 0, stream, a, fw, rc);
        } else if (a.k == 15) hipLaunchKernelGGL((verify_scan_kernel<15, false>), grid, dim3(VS_THREADS), 0, stream, a, fw, rc);
        else hipLaunchKernelGGL((verify_scan_kernel<0, false>), grid, dim3(VS_THREADS), 0, stream, a, fw, rc);
    } else {
        hipLaunchKernelGGL(cand_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, fw);
        hipLaunchKernelGGL(cand_gather_kernel, dim3(fw.n_slices), dim3(64), 0, stream, fw);
#ifdef DRPRG_EXPERIMENTAL
        if (by_read) {
            fw.verify_grid = std::min<uint32_t>(read_verify_grid(n_cus), MAX_EX_WG);
            HIP_TRY(launch_read_verify(a, fw, rc, fw.verify_grid, stream));
        } else
#endif
        if (a.packed) {
            if (a.k == 15) hipLaunchKernelGGL((verify_count_kernel<15, true>), dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw, rc);
            else hipLaunchKernelGGL((verify_count_kernel<0, true>), dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw, rc);
        } else if (a.k == 15) hipLaunchKernelGGL((verify_count_kernel<15, false>), dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw, rc);
        else hipLaunchKernelGGL((verify_count_kernel<0, false>), dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw, rc);
    }
    if (with_totals) hipLaunchKernelGGL(hit_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, a, fw, 0, fw.verify_grid); // (else: read_cluster_kernel's workgroup 0)
    return hipGetLastError();
}

// DRPRG_DIRECT_FORM=lds keeps sketch_probe_kernel for every (k, w) (A/B runs, and a second way through the parity tests)
static bool use_wave_form(int k, int w, bool wide_hash)
{
    static const bool forced_lds = [] {
        const char* e = std::getenv("DRPRG_DIRECT_FORM");
        return e && std::string(e) == "lds";
    }();
    return !wide_hash && !forced_lds && wave_kernel_applies(k, w);
}

bool direct_uses_wave_form(int k, int w, bool wide_hash) { return use_wave_form(k, w, wide_hash); }

uint32_t direct_candidate_tiles(uint64_t n_bases, int halo, int k, int w, bool wide_hash)
{
    return use_wave_form(k, w, wide_hash) ? wave_n_slices(n_bases) : sketch_n_tiles(n_bases, halo);
}

uint32_t direct_first_read_tiles(uint64_t n_bases, int halo, int k, int w, bool wide_hash)
{
    return use_wave_form(k, w, wide_hash) ? wave_n_tiles(n_bases) : sketch_n_tiles(n_bases, halo);
}

hipError_t launch_direct_candidates(const SketchArgs& a, bool wide_hash, uint32_t* tile_prefix, void* temp, size_t temp_bytes,
    uint64_t dense_capacity, const ReadClusterArgs& rc, int n_cus, FilterWork& fw, hipStream_t stream, KernelTimer timer, uint32_t slices_mark)
{
    if (a.n_bases == 0 || !a.tile_cap) return hipErrorInvalidValue;
    const uint32_t n_tiles = direct_candidate_tiles(a.n_bases, a.halo, a.k, a.w, wide_hash);
    if (use_wave_form(a.k, a.w, wide_hash)) HIP_TRY(launch_sketch_wave(a, stream, timer));
    else HIP_TRY(launch_sketch_probe(a, wide_hash, stream, timer));
    // tile_count[n_tiles] is a zero the caller keeps there: the exclusive scan of n_tiles + 1 counts ends with the total
    HIP_TRY(exclusive_scan_u32(temp, temp_bytes, a.tile_count, tile_prefix, n_tiles + 1, stream));
    fw.cand_total = tile_prefix + n_tiles;
    const dim3 grid((n_tiles + TG_TILES - 1) / TG_TILES);
    // (the block table of the slices form lives in the dense cand_info array: two words per entry of capacity, one per 64 needed)
    uint32_t* const block_first = slices_mark && dense_capacity ? reinterpret_cast<uint32_t*>(fw.cand_info) : nullptr;
    hipLaunchKernelGGL(tile_totals_kernel, dim3(std::min<uint32_t>((n_tiles + TG_THREADS - 1) / TG_THREADS, (uint32_t)n_cus)), dim3(TG_THREADS), 0, stream, a,
        tile_prefix, n_tiles, dense_capacity, block_first);
    HIP_TRY(hipGetLastError());
    if (slices_mark) { // read_cluster_kernel takes the candidates from the slices: a dense list only if reads are left over
        ReadClusterArgs rcs = rc;
        rcs.block_first = block_first;
        rcs.slice_prefix = tile_prefix;
        rcs.n_slices = n_tiles;
        rcs.mark_epoch = slices_mark;
        return launch_read_cluster(a, fw, rcs, n_cus, false, stream);
    }
    hipLaunchKernelGGL(tile_gather_kernel, grid, dim3(TG_THREADS), 0, stream, a, fw, tile_prefix, n_tiles, dense_capacity, 0u);
    HIP_TRY(hipGetLastError());
    return launch_read_cluster(a, fw, rc, n_cus, false, stream);
}

// the dense list after all, for the reads read_cluster_kernel<SLICES> left over: the slices copied, handled candidates (dense
// cand_pos1 == mark) with position 0
hipError_t launch_tile_gather_marked(const SketchArgs& a, const FilterWork& fw, const uint32_t* tile_prefix, uint32_t n_tiles, uint64_t dense_capacity,
    uint32_t mark, hipStream_t stream)
{
    hipLaunchKernelGGL(tile_gather_kernel, dim3((n_tiles + TG_TILES - 1) / TG_TILES), dim3(TG_THREADS), 0, stream, a, fw, tile_prefix, n_tiles,
        dense_capacity, mark);
    return hipGetLastError();
}

hipError_t launch_filter_recount(const SketchArgs& a, const FilterWork& fw, hipStream_t stream)
{
    hipLaunchKernelGGL(recount_kernel, dim3(fw.ex_grid), dim3(EX_THREADS), 0, stream, a, fw);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(hit_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, a, fw, 1, fw.ex_grid);
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
