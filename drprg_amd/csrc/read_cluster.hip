// read_cluster.hip -- the last stage of the filtered launch sequence (see the overview at the top of
// sketch_filter.hip): clusters, size / overlap filters and coverage straight from the ordered candidate list.
#include "filter_common.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace drprg {
namespace dev {

// ---------------------------------------------------------------------------------------------
// per-read clustering straight from the candidate list
// ---------------------------------------------------------------------------------------------
// The candidates arrive ordered by (read, position) -- from verify_count_kernel, or from the direct sketch kernel in its
// candidate form -- so all minimizers of a read sit next to each other, and a short read has a few dozen hits, nearly
// always in ONE cluster.  read_cluster_kernel therefore never materialises the hit list.  A workgroup stages RC_SLOTS
// consecutive candidates and their index records (the hits) in LDS.  A position gap > max_diff between two consecutive
// minimizers of a read starts a new segment; as long as all hits of the read lie in one (prg, strand) group, the
// segments ARE the clusters and the overlap sweep of cluster_filter_kernel cannot drop any of them (same group,
// disjoint position ranges).  So: a segment's hit count is a difference of two scan values, the first slot of a segment
// compares it with the size threshold of cluster_eval_kernel (stored per candidate), and every minimizer of a kept
// segment then adds its own hits to the coverage vector -- all of it data parallel.  Only the reads whose hits fall into
// several groups are queued and handled one wave per read: lane j holds cluster j, the hits are broadcast one by one
// (clusters per group split at gaps, size threshold, rank in clusterComp order, the overlap sweep; pandora
// define_clusters / filter_clusters).  This replaces expand + reorder + flag + scan + start + eval + filter + count +
// accumulate (13 launches) for such reads.  A read that does not fit (its candidates run past the staged range, more
// than 64 clusters, more than RC_HCAP staged hits in the chunk, a position >= 2^16 - 2, more than RC_POOL multi-group
// reads in a chunk) is left alone: its candidates keep cand_pos1 != 0, n_complex counts it, and the host sends what is
// left through the generic pipeline.  Handled reads get cand_pos1 = 0.  (A workgroup may read cand_pos1 of a
// neighbouring chunk's read while that chunk zeroes it: either value only moves where the foreign hits land in LDS,
// nothing else.)  Chunks are handed out through a global counter.
constexpr int RC_THREADS = 1024;
constexpr int RC_WAVES = RC_THREADS / 64;
#ifndef DRPRG_RC_PER // (build-time knobs of tools/rc_variants.sh: slots per thread, staged hits, workgroups per CU)
#define DRPRG_RC_PER 2
#define DRPRG_RC_HCAP 3072
#define DRPRG_RC_WG_PER_CU 2
#endif
constexpr int RC_PER = DRPRG_RC_PER;
constexpr int RC_SLOTS = RC_THREADS * RC_PER; // staged candidates
// Look-ahead (template parameter AHEAD; RC_AHEAD / RC_OWN are defined at the top of the kernel): a read belongs to the chunk that owns its
// first candidate, so the last AHEAD of the staged slots are there for reads that begin in the owned range and run on.  512 slots serve
// a 4 kb Nanopore read (~250 candidates); a 150-base read has a few dozen at most, and every slot not spent on look-ahead is owned:
// 128 slots of look-ahead mean 1920 owned candidates per chunk instead of 1536, a fifth fewer chunks -- and a chunk costs ~21 us whatever
// is in it (round 4).  A read that does not fit its chunk's look-ahead goes through the generic pipeline as before.
constexpr int RC_HCAP = DRPRG_RC_HCAP;         // staged hits
constexpr int RC_POOL = 512;                  // reads per chunk that may take the wave path (sketch_wave_kernel clusters the plain reads itself: what is left is rich in these)
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
// the same, and the thread index comes out of it as a value the compiler knows nothing about: the LDS addresses derived from
// it are computed again after every barrier instead of being kept in registers across the whole chunk loop (at 64 VGPRs --
// two 1024-thread workgroups per CU -- the kernel spilled twelve of them to scratch memory)
__device__ __forceinline__ void lds_barrier(int& t) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(t) : : "memory"); }

// SLICES: the candidates are read from the tile slices of the direct sketch kernel (rc.slice_prefix), not from a gathered list
template <bool SLICES, int AHEAD>
__global__ __launch_bounds__(RC_THREADS, 4 * DRPRG_RC_WG_PER_CU) void read_cluster_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    constexpr int RC_AHEAD = AHEAD, RC_OWN = RC_SLOTS - AHEAD;
    static_assert(RC_PER != 2 || AHEAD != 512 || RC_OWN == (int)RC_CHUNK_OWN, "kernels.h RC_CHUNK_OWN: the chunk numbering the wave form's flags use");
    static_assert(RC_OWN % 64 == 0, "the slices form locates 64 entries from a multiple of 64 per wave");
    (void)RC_AHEAD;
    extern __shared__ uint32_t s_hist[]; // clusters kept per PRG
    __shared__ uint32_t s_read[RC_SLOTS + 1], s_hstart[RC_SLOTS + 1];
    __shared__ uint16_t s_pos1[RC_SLOTS]; // read position + 1 of a minimizer (0xFFFF: too far for this kernel), 0 = not a minimizer
    __shared__ uint32_t s_gt[RC_SLOTS];   // at a segment's first slot: group << 16 | size threshold
    __shared__ uint16_t s_lead[RC_SLOTS]; // 1 + slot of the first candidate of this slot's read (0: the read started in an earlier chunk)
    __shared__ uint16_t s_seg[RC_SLOTS];  // 1 + first slot of this slot's segment
    __shared__ uint16_t s_end[RC_SLOTS];  // at a segment's first slot: the first slot of the next segment
    __shared__ uint16_t s_g0[RC_SLOTS];   // group (prg << 1 | rev) of the slot's first hit
    __shared__ uint8_t s_dec[RC_SLOTS];   // at a segment's first slot: 0 leave alone, 1 handled, 2 handled and every hit counts
    __shared__ uint8_t s_cplx[RC_SLOTS], s_irrf[RC_SLOTS]; // at a read's first slot: does not fit / hits in several groups
    __shared__ uint16_t s_grp[RC_HCAP];   // per hit: prg << 1 | rev
    __shared__ uint16_t s_hpos[RC_HCAP];  // per hit: read position
    __shared__ uint32_t s_cov[RC_HCAP];   // per hit: index into the coverage vector, 2 * k-mer node + rev
    __shared__ uint32_t s_w[2][RC_WAVES];
    __shared__ uint32_t s_irr[RC_POOL];
    __shared__ uint32_t s_prev_read, s_n_irr, s_chunk;
    __shared__ unsigned long long s_tot[3];

    int tid = threadIdx.x;
#define lane (tid & 63)
#define wave (tid >> 6)
    // (DRPRG_RC_DEBUG=1: thread 0 of every workgroup adds the clock cycles between two marks to counter `ph`)
    unsigned long long pc_last = rc.phase_clock ? clock64() : 0;
#define RC_MARK(ph)                                                  \
    do {                                                             \
        if (rc.phase_clock && tid == 0) {                            \
            const unsigned long long pc_now = clock64();             \
            atomicAdd(&rc.phase_clock[ph], pc_now - pc_last);        \
            pc_last = pc_now;                                        \
        }                                                            \
    } while (0)
    if (rc.n_wg && blockIdx.x == 0) { // the candidate stage's totals (kernels.h): before any of the early returns below
        uint32_t h = 0, nm = 0, ml = 0;
        for (uint32_t g = (uint32_t)tid; g < rc.n_wg; g += RC_THREADS) {
            h += rc.wg_hits[g];
            nm += rc.wg_nmin[g];
            const uint32_t m = rc.wg_maxlen[g];
            ml = m > ml ? m : ml;
        }
        const uint32_t wh = wave_inclusive_scan(h), wn = wave_inclusive_scan(nm), wm = wave_max(ml);
        if (lane == 63) {
            s_w[0][wave] = wh;
            s_w[1][wave] = wn;
            s_irr[wave] = wm;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long th = 0, tn = 0;
            uint32_t tm = 0;
            for (int i = 0; i < RC_WAVES; ++i) {
                th += s_w[0][i];
                tn += s_w[1][i];
                tm = s_irr[i] > tm ? s_irr[i] : tm;
            }
            *rc.tot_hits = th;
            if (tn && !(*reinterpret_cast<volatile const uint32_t*>(rc.overflow_word) & 4u)) atomicAdd(rc.tot_minimizers, tn);
            *rc.tot_max_len = (unsigned long long)tm;
        }
        __syncthreads();
    }
    if (*reinterpret_cast<volatile uint32_t*>(a.overflow) & 4u) return; // a candidate slice overflowed: the host re-runs the batch
    // second pass behind read_cluster_wave_kernel: only if that kernel left reads untouched (long reads, mostly)
    if (rc.second_pass && *reinterpret_cast<volatile unsigned long long*>(rc.n_unfit) == 0ull) return;
    const uint32_t total = SLICES ? rc.slice_prefix[rc.n_slices] : *fw.cand_total;
    const uint32_t handled_mark = SLICES ? rc.mark_epoch : 0u; // what a handled candidate's cand_pos1 becomes
    // (the shortest path of every PRG sits behind the histogram when two workgroups per CU still fit with it -- launch_read_cluster decides:
    // the wave path reads it per cluster, and a global load there is a round trip on one wave with the rest of the workgroup waiting at
    // the next barrier.  16 bits each; 0xFFFF = look it up)
    uint16_t* const s_minpath = reinterpret_cast<uint16_t*>(s_hist + rc.n_prgs);
    for (uint32_t i = tid; i < rc.n_prgs; i += RC_THREADS) {
        s_hist[i] = 0;
        if (rc.minpath_in_lds) {
            const uint32_t m = rc.prg_min_path_len[i];
            s_minpath[i] = (uint16_t)(m < 0xFFFFu ? m : 0xFFFFu);
        }
    }
    if (tid < 3) s_tot[tid] = 0;
    unsigned long long my_kept_hits = 0;
    uint32_t my_kept = 0, my_complex = 0;
    const uint32_t w1_magic = w1_reciprocal(a.w);

    // chunks are handed out by a global counter (two or three chunks per workgroup: a static split leaves a third of
    // the workgroups idle for the last round); the next chunk number is fetched while the current one is processed, and
    // the next chunk's candidates are requested before the wave path (G) of the current one: that path keeps one or two waves
    // busy for microseconds while the others wait, which is when the loads travel
    struct Staged { // the raw loads of one chunk: slot tid + q * RC_THREADS, and the two neighbours of the staged range
        uint4 rec[RC_PER];
        uint32_t info[RC_PER], pos1[RC_PER], edge_prev, edge_next;
    };
    auto request = [&](uint32_t chunk) {
        Staged st;
        const uint64_t b64 = (uint64_t)chunk * RC_OWN;
        const bool live = b64 < total; // (past the end: nothing is read, the loop ends on this chunk)
        const uint32_t b = live ? (uint32_t)b64 : 0u;
        const uint32_t n_in = total - b < (uint32_t)RC_SLOTS ? total - b : (uint32_t)RC_SLOTS;
        if constexpr (!SLICES) {
#pragma unroll
            for (int q = 0; q < RC_PER; ++q) { // (all six loads together and without a branch: a slot past the end reads the chunk's first
                                               // candidate and drops it)
                const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
                const uint32_t src = b + (i < n_in ? i : 0u);
                st.info[q] = live ? (uint32_t)fw.cand_info[src] : 0u;
                st.pos1[q] = live ? fw.cand_pos1[src] : 0u;
                st.rec[q] = live ? fw.cand_rec[src] : make_uint4(0, 0, 0, 0);
            }
            const bool nxt = n_in == (uint32_t)RC_SLOTS && b + n_in < total;
            st.edge_prev = live ? (uint32_t)fw.cand_info[b ? b - 1 : 0u] : 0u;
            st.edge_next = live ? (uint32_t)fw.cand_info[nxt ? b + n_in : b] : 0u;
        } else {
            // Where entry d of the ordered list lives in the slice arrays.  rc.block_first[m] is the slice that holds entry 64 m
            // (written by tile_totals_kernel), a wave's 64 slots are 64 consecutive entries from a multiple of 64, and the starts of
            // the 64 slices behind that one are one coalesced load: every start inside the wave's range is scattered to the wave's own
            // 64 words of LDS as 1 + its lane (the last slice of a run of equal starts -- empty slices -- wins) and an inclusive max
            // scan over the lanes gives each slot its slice.  No workgroup barrier, two dependent loads where the round-3 form
            // bisected a window of the prefix in LDS (eleven dependent LDS reads per slot: a quarter of a chunk's time).  If more
            // than 64 slices start inside the range (a batch with next to no candidates), every slot bisects the prefix in global memory.
            // The loads go out in two rounds for all of a wave's ranges together -- the block table words (scalar loads), then the
            // slice starts -- so that a request costs two memory latencies, not two per range.
            const uint32_t* __restrict__ P = rc.slice_prefix;
            uint32_t* const s_mine = s_gt + (wave << 6); // (s_gt is free between phase E and the next phase C)
            const uint32_t d_last = b + n_in - 1; // (n_in >= 1 when live)
            const bool nxt = n_in == (uint32_t)RC_SLOTS && b + n_in < total;
            // ranges: q < RC_PER the wave's own slots; RC_PER: the 64 entries before b (wave 0: entry b - 1 is its slot 63)
            uint32_t d0[RC_PER + 1], s0[RC_PER + 1], pj[RC_PER + 1], p0[RC_PER + 1], sn = 0, pn0 = 0;
            bool some[RC_PER + 1];
#pragma unroll
            for (int q = 0; q <= RC_PER; ++q) {
                d0[q] = q < RC_PER ? b + ((uint32_t)(wave << 6) + (uint32_t)q * RC_THREADS) : b - 64u;
                some[q] = q < RC_PER ? live && d0[q] <= d_last : live && b && wave == 0; // (wave-uniform)
                s0[q] = some[q] ? rc.block_first[__builtin_amdgcn_readfirstlane(d0[q] >> 6)] : 0u;
            }
            const bool edge_n = live && nxt && wave == 0; // entry b + RC_SLOTS is the first of its 64: in the slice the table names
            if (edge_n) sn = rc.block_first[__builtin_amdgcn_readfirstlane((b + n_in) >> 6)];
#pragma unroll
            for (int q = 0; q <= RC_PER; ++q) {
                s0[q] = __builtin_amdgcn_readfirstlane(s0[q]);
                const uint32_t j1 = s0[q] + 1u + (uint32_t)lane;
                pj[q] = some[q] ? P[j1 < rc.n_slices ? j1 : rc.n_slices] : 0u;
                p0[q] = some[q] ? P[s0[q]] : 0u;
            }
            if (edge_n) pn0 = P[sn];
            auto locate = [&](int q, uint32_t d_hi) -> size_t { // entry min(d0 + lane, d_hi); d0 a multiple of 64, d0 <= d_hi < total
                const uint32_t x = d0[q] + (uint32_t)lane <= d_hi ? (uint32_t)lane : d_hi - d0[q];
                if (__builtin_amdgcn_readlane(pj[q], 63) > d_hi) {
                    s_mine[lane] = 0;
                    const uint32_t pn = __shfl_down(pj[q], 1);
                    const uint32_t rel = pj[q] - d0[q];
                    if ((lane == 63 || pn != pj[q]) && rel < 64u) s_mine[rel] = (uint32_t)lane + 1u;
                    __builtin_amdgcn_wave_barrier();
                    uint32_t c = s_mine[lane];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t n = __shfl_up(c, off);
                        if (lane >= off) c = n > c ? n : c;
                    }
                    c = __shfl(c, (int)x);
                    const uint32_t before = __shfl(pj[q], (int)(c ? c - 1u : 0u));
                    return (size_t)(s0[q] + c) * a.tile_cap + (d0[q] + x - (c ? before : p0[q]));
                }
                const uint32_t d = d0[q] + x;
                uint32_t lo = s0[q], hi = rc.n_slices; // P[lo] <= d < P[hi]
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (P[mid] <= d) lo = mid;
                    else hi = mid;
                }
                return (size_t)lo * a.tile_cap + (d - P[lo]);
            };
#pragma unroll
            for (int q = 0; q < RC_PER; ++q) {
                const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
                const size_t src = some[q] ? locate(q, d_last) : 0; // (a wave past the end reads entry 0 of slice 0 and drops it)
                st.info[q] = live ? (uint32_t)a.tile_info[src] : 0u;
                st.pos1[q] = live ? a.tile_pos1[src] : 0u;
                st.rec[q] = live ? a.tile_rec[src] : make_uint4(0, 0, 0, 0);
                // (second pass: what the wave form handled carries this batch's mark in the dense array)
                if (rc.second_pass && live && i < n_in && fw.cand_pos1[b + i] == rc.mark_epoch) st.pos1[q] = 0u;
            }
            // the two neighbours of the staged range (thread 0 alone looks at them)
            st.edge_prev = 0u;
            st.edge_next = 0u;
            if (wave == 0) {
                size_t e_prev = 0;
                if (some[RC_PER]) e_prev = (size_t)__shfl((unsigned long long)locate(RC_PER, b - 1u), 63);
                const size_t e_next = edge_n ? (size_t)sn * a.tile_cap + (b + n_in - pn0) : 0;
                st.edge_prev = some[RC_PER] ? (uint32_t)a.tile_info[e_prev] : 0u;
                st.edge_next = edge_n ? (uint32_t)a.tile_info[e_next] : 0u;
            }
        }
        return st;
    };
    // (second pass: only the chunks in which the wave form left a read)
    // Chunks 0 .. gridDim.x - 1 belong to the workgroups by number, the counter hands out the ones after them: every workgroup's first
    // ticket used to be a returning atomic on one address, 512 of them at the same moment -- microseconds in which nothing else of the
    // workgroup could start (round 5; the fixed cost of a launch went from 21 to 17 us)
    auto take_ticket = [&]() -> uint32_t {
        uint32_t t;
        do t = atomicAdd(rc.chunk_counter, 1u) + gridDim.x;
        while (rc.second_pass && (uint64_t)t * RC_OWN < total && !rc.chunk_flags[t]);
        return t;
    };
    uint32_t cur = blockIdx.x;
    if (rc.second_pass) { // (experimental wave form first: a chunk of its own only if that form left a read in it)
        if (tid == 0) s_chunk = ((uint64_t)cur * RC_OWN < total && !rc.chunk_flags[cur]) ? take_ticket() : cur;
        lds_barrier(tid);
        cur = s_chunk;
    }
    Staged nx = request(cur);
    for (;;) {
        RC_MARK(9); // (what ran since mark 8: the wave path of thread 0's wave)
        lds_barrier(tid); // LDS of the previous chunk is free, everybody holds the chunk number in a register
        RC_MARK(0);
        const uint64_t base64 = (uint64_t)cur * RC_OWN;
        if (base64 >= total) break;
        const uint32_t base = (uint32_t)base64;
        const uint32_t n_loaded = total - base < (uint32_t)RC_SLOTS ? total - base : (uint32_t)RC_SLOTS;
        const uint32_t n_own = total - base < (uint32_t)RC_OWN ? total - base : (uint32_t)RC_OWN;
        RC_MARK(1);
        // (the ticket of the next chunk: a device-wide atomic that returns a value takes microseconds; it is asked for here and put
        // into LDS two phases later, so that nobody waits for it at the barrier that closes stage A)
        uint32_t ticket = 0;
        if (tid == 0) ticket = take_ticket();
        // ---- A: stage the candidates (requested one chunk ago) ----
        uint4 crec[RC_PER];
        uint32_t st_read[RC_PER], st_pos1[RC_PER];
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            st_read[q] = nx.info[q];
            st_pos1[q] = nx.pos1[q];
            crec[q] = nx.rec[q];
        }
        const bool has_next = n_loaded == (uint32_t)RC_SLOTS && base + n_loaded < total;
        const uint32_t edge_prev = nx.edge_prev, edge_next = nx.edge_next;
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const bool in = i < n_loaded;
            st_read[q] = in ? st_read[q] & 0x7FFFFFFFu : READ_NONE;
            st_pos1[q] = in ? st_pos1[q] : 0u;
            if (!in) crec[q] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const uint32_t read = st_read[q], pos1 = st_pos1[q];
            if (!pos1) crec[q].y = 0; // handled by another chunk in the meantime (look-ahead slots only)
            s_read[i] = read;
            s_pos1[i] = (uint16_t)(pos1 < 0xFFFFu ? pos1 : 0xFFFFu);
            s_hstart[i] = crec[q].y;
            s_g0[i] = (uint16_t)((crec[q].z >> 16) & 0x7FFFu);
            s_cplx[i] = 0;
            s_irrf[i] = 0;
        }
        if (tid == 0) {
            s_prev_read = base ? (edge_prev & 0x7FFFFFFFu) : 0xFFFFFFFFu;
            // the candidate after the staged range: a read that runs on into it does not fit
            s_read[RC_SLOTS] = has_next ? (edge_next & 0x7FFFFFFFu) : 0xFFFFFFFFu;
            s_n_irr = 0;
        }
        lds_barrier(tid);
        RC_MARK(2);
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
            lds_barrier(tid);
            RC_MARK(3);
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
        if (tid == 0) s_chunk = ticket;
        lds_barrier(tid);
        RC_MARK(4);
        // ---- C: the hits (one per index record of every minimizer); every segment start closes the segment before it; a
        // minimizer whose group differs from the previous one of its read makes the read irregular; the first minimizer of a
        // segment names the segment's group and threshold ----
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const uint32_t seg1 = s_seg[i];
            if (seg1 == i + 1 && i > 0 && s_seg[i - 1]) s_end[s_seg[i - 1] - 1] = (uint16_t)i;
            if (i == RC_SLOTS - 1 && seg1) s_end[seg1 - 1] = (uint16_t)RC_SLOTS;
            const uint32_t cnt = crec[q].y;
            if (!cnt) continue;
            const uint32_t h0 = s_hstart[i], lead = s_lead[i];
            const bool mine = lead != 0 && lead <= n_own; // the read starts in the owned range
            const uint32_t pos = (uint32_t)s_pos1[i] - 1, strand = crec[q].z >> 31, g = (crec[q].z >> 16) & 0x7FFFu;
            if (h0 + cnt > (uint32_t)RC_HCAP || pos >= 0xFFFEu) {
                if (mine) s_cplx[lead - 1] = 1;
                continue;
            }
            bool irregular = false;
            s_grp[h0] = (uint16_t)g;
            s_hpos[h0] = (uint16_t)pos;
            s_cov[h0] = crec[q].w;
            // a k-mer that several k-mer nodes share (a quarter of the minimizers of a PRG index): its further records, four at a
            // time with their eight loads requested together (one after the other they were up to three round trips that the
            // whole workgroup waited for at the barrier)
            if (fw.debug & 4096u) // (DRPRG_FT_DEBUG=4096: timing only, wrong results -- what fetching the further records costs: they repeat the first)
                for (uint32_t r = 1; r < cnt; ++r) {
                    s_grp[h0 + r] = (uint16_t)g;
                    s_hpos[h0 + r] = (uint16_t)pos;
                    s_cov[h0 + r] = crec[q].w;
                }
            for (uint32_t r0 = 1; r0 < cnt && !(fw.debug & 4096u); r0 += 4) {
                uint32_t kn4[4], prg4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t r = r0 + (uint32_t)u < cnt ? r0 + (uint32_t)u : r0;
                    kn4[u] = a.rec_knode[crec[q].x + r];
                    prg4[u] = a.rec_prg[crec[q].x + r];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t r = r0 + (uint32_t)u;
                    if (r >= cnt) break;
                    const uint32_t rev = ((kn4[u] & 1u) == strand) ? 0u : 1u;
                    irregular |= ((prg4[u] << 1) | rev) != g;
                    s_grp[h0 + r] = (uint16_t)((prg4[u] << 1) | rev);
                    s_hpos[h0 + r] = (uint16_t)pos;
                    s_cov[h0 + r] = (kn4[u] >> 1) * 2u + rev;
                }
            }
            if (!mine) continue;
            const uint32_t first = lead - 1, seg = seg1 - 1;
            int j = (int)i - 1; // the previous minimizer of the read
            while (j >= (int)first && !s_pos1[j]) --j;
            if (j >= (int)first && s_g0[j] != g) irregular = true;
            if (j < (int)seg) s_gt[seg] = crec[q].z & 0x7FFFFFFFu;
            if (irregular) s_irrf[first] = 1;
        }
        if (tid == 0 && s_read[RC_SLOTS] == s_read[RC_SLOTS - 1]) { // the last staged read runs on past the staged range
            const uint32_t lead = s_lead[RC_SLOTS - 1];
            if (lead && lead <= n_own) s_cplx[lead - 1] = 1;
        }
        lds_barrier(tid);
        RC_MARK(5);
        // ---- E: the first slot of every segment decides for the segment; reads with several groups queue for the wave path ----
#pragma unroll
        for (int q = 0; q < RC_PER; ++q) {
            const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
            const uint32_t lead = s_lead[i];
            if (i >= n_loaded || s_seg[i] != i + 1 || lead == 0 || lead > n_own) continue;
            uint32_t flags = (s_cplx[lead - 1] ? RC_COMPLEX : 0u) | (s_irrf[lead - 1] ? RC_IRREGULAR : 0u);
            // A read with hits in several groups whose hits, all groups together, do not exceed the smallest size threshold a
            // cluster can have cannot keep any cluster: it is done without the wave path (every segment of the read sees the same
            // total, so all of them decide alike).  Most such reads are of this kind: a stray hit or two in another group.
            bool dropped = false;
            uint32_t e = i; // the first slot after the read: follow its segments
            if (flags == RC_IRREGULAR) {
                for (uint32_t sg = i; sg < (uint32_t)RC_SLOTS && s_lead[sg] == lead; sg = s_end[sg]) e = s_end[sg];
                dropped = s_hstart[e] - s_hstart[lead - 1] <= rc.min_cluster_size;
            }
            if (lead == i + 1) { // first slot of the read
                if (flags & RC_COMPLEX) ++my_complex;
                else if ((flags & RC_IRREGULAR) && !dropped) {
                    const uint32_t at = atomicAdd(&s_n_irr, 1u);
                    if (at < (uint32_t)RC_POOL) s_irr[at] = i | (e << 16);
                    else ++my_complex;
                }
            }
            const uint32_t n_hits = s_hstart[s_end[i]] - s_hstart[i];
            uint32_t dec = dropped ? 1u : 0u; // 0 leave alone, 1 handled, 2 handled and every hit counts
            if (!flags && n_hits) {
                const uint32_t g_thr = s_gt[i];
                dec = 1;
                if (n_hits > (g_thr & 0xFFFFu)) {
                    dec = 2;
                    atomicAdd(&s_hist[g_thr >> 17], 1u);
                    ++my_kept;
                    my_kept_hits += n_hits;
                }
            }
            s_dec[i] = (uint8_t)dec;
        }
        lds_barrier(tid);
        RC_MARK(6);
        // ---- F: the minimizers of the kept segments ----
        {
#pragma unroll
            for (int q = 0; q < RC_PER; ++q) {
                const uint32_t i = (uint32_t)tid + (uint32_t)q * RC_THREADS;
                const uint32_t lead = s_lead[i], cnt = crec[q].y;
                if (!cnt || lead == 0 || lead > n_own) continue;
                const uint32_t decision = s_dec[(uint32_t)s_seg[i] - 1];
                if (!decision) continue;
                fw.cand_pos1[base + i] = handled_mark; // handled
                if (decision == 2) {
                    atomicAdd(&rc.covg[crec[q].w], 1u);
                    const uint32_t h0 = s_hstart[i];
                    for (uint32_t r = 1; r < cnt; ++r) atomicAdd(&rc.covg[s_cov[h0 + r]], 1u);
                }
            }
        }
        RC_MARK(8);
        cur = s_chunk; // (written by thread 0 before the barrier that closed phase B)
        nx = request(cur);
        RC_MARK(12); // (asking for the next chunk's candidates; what follows until mark 9 is the wave path of thread 0's wave)
        // ---- G: reads with hits in several groups, one wave per read: lane j holds cluster j, the hits are broadcast one by one
        // (clusters per group split at gaps, size threshold, the overlap sweep of cluster_filter_kernel) ----
        const uint32_t n_irr = s_n_irr < (uint32_t)RC_POOL ? s_n_irr : (uint32_t)RC_POOL;
        if (rc.phase_clock && tid == 0) { // (counters 10, 11 of the debug block: reads that took the wave path, chunks)
            atomicAdd(&rc.phase_clock[10], (unsigned long long)n_irr);
            atomicAdd(&rc.phase_clock[11], 1ull);
        }
        for (uint32_t r = wave; r < n_irr; r += RC_WAVES) {
            const uint32_t i = s_irr[r] & 0xFFFFu, e = s_irr[r] >> 16;
            const uint32_t read = s_read[i], hb = s_hstart[i], he = s_hstart[e];
            const uint64_t len = a.offsets[read + 1] - a.offsets[read]; // (requested here, needed after the loop over the hits)
            uint32_t cl_g = 0, cl_n = 0, cl_first = 0, cl_last = 0;
            int nc = 0;
            bool complex = false;
            for (uint32_t b = hb; b < he && !complex; b += 64) {
                const uint32_t h = b + lane;
                const uint32_t hg = h < he ? s_grp[h] : 0u, hp = h < he ? s_hpos[h] : 0u;
                // The hits of the batch, in position order, one group after the other (a read has two or three): within a group a
                // hit starts a new cluster iff the previous hit of its group lies more than max_diff before it -- for the first one:
                // iff there is no cluster of the group yet, or its last hit lies that far back -- so the starts come out of one
                // ballot, every new cluster lane picks its span out of the group's mask, and the hits before the first start extend
                // the group's latest cluster.  (One hit at a time -- find the group's latest cluster, compare, join or create -- was
                // a chain of two hundred cycles per hit, on one wave, with the rest of the workgroup waiting at the next barrier.)
                const bool valid = h < he;
                const uint64_t lt = (1ull << lane) - 1ull;
                uint64_t todo = __ballot(valid);
                while (todo) {
                    const uint32_t G = (uint32_t)__builtin_amdgcn_readlane((int)hg, __ffsll((long long)todo) - 1);
                    const uint64_t m = __ballot(valid && hg == G);
                    todo &= ~m;
                    const uint64_t open = __ballot(lane < nc && cl_g == G); // the group's clusters so far: the highest lane is the latest
                    const int fo = open ? 63 - __clzll((long long)open) : -1;
                    const uint32_t open_last = fo >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)cl_last, fo) : 0u;
                    const uint64_t below = m & lt;
                    const int prev = below ? 63 - __clzll((long long)below) : -1;
                    const uint32_t prev_pos = __shfl(hp, prev < 0 ? lane : prev); // (every lane takes part in the shuffle)
                    const bool member = ((m >> lane) & 1ull) != 0;
                    const bool start = member
                        && (prev >= 0 ? (int)(hp - prev_pos) > rc.max_diff : (fo < 0 || (int)(hp - open_last) > rc.max_diff));
                    const uint64_t starts = __ballot(start);
                    const int n_new = __popcll(starts);
                    if (nc + n_new > 64) {
                        complex = true;
                        break;
                    }
                    const uint64_t cont = m & (starts ? (1ull << (__ffsll((long long)starts) - 1)) - 1ull : ~0ull); // before the first start
                    if (cont) { // (then the group has a cluster: otherwise its first hit would be a start)
                        const uint32_t lp = (uint32_t)__builtin_amdgcn_readlane((int)hp, 63 - __clzll((long long)cont));
                        if (lane == fo) {
                            cl_n += (uint32_t)__popcll(cont);
                            cl_last = lp;
                        }
                    }
                    const int c = lane - nc; // lane nc + c takes the c-th start
                    const bool mine = c >= 0 && c < n_new;
                    int first_lane = lane, last_lane = lane;
                    uint32_t span_n = 0;
                    if (mine) {
                        uint64_t mm = starts;
                        for (int k = 0; k < c; ++k) mm &= mm - 1;
                        first_lane = __ffsll((long long)mm) - 1;
                        const uint64_t rest = mm & (mm - 1);
                        const uint64_t upto = rest ? (1ull << (__ffsll((long long)rest) - 1)) - 1ull : ~0ull; // lanes below the next start
                        const uint64_t span = m & upto & ~((1ull << first_lane) - 1ull);
                        last_lane = 63 - __clzll((long long)span);
                        span_n = (uint32_t)__popcll(span);
                    }
                    const uint32_t fpos = __shfl(hp, first_lane), lpos = __shfl(hp, last_lane);
                    if (mine) {
                        cl_g = G;
                        cl_n = span_n;
                        cl_first = fpos;
                        cl_last = lpos;
                    }
                    nc += n_new;
                }
            }
            if (complex) {
                if (lane == 0) ++my_complex;
                continue;
            }
            const uint64_t expected = expected_minimizers(len, a.w, w1_magic); // (2 len / (w + 1) without a 64-bit division)
            bool kept = false;
            if (lane < nc) {
                uint64_t m = rc.minpath_in_lds ? s_minpath[cl_g >> 1] : 0xFFFFu;
                if (m == 0xFFFFu) m = rc.prg_min_path_len[cl_g >> 1];
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
                    const uint32_t of = (uint32_t)__builtin_amdgcn_readlane((int)cl_first, o), on = (uint32_t)__builtin_amdgcn_readlane((int)cl_n, o),
                                   og = (uint32_t)__builtin_amdgcn_readlane((int)cl_g, o);
                    rank += (of < cl_first || (of == cl_first && (on > cl_n || (on == cl_n && og < cl_g)))) ? 1u : 0u;
                }
                const int nk = __popcll(alive);
                int prev = -1;
                uint32_t pg = 0, pn = 0, p_last = 0;
                for (int o = 0; o < nk; ++o) {
                    const int cur = __ffsll((long long)__ballot(kept && rank == (uint32_t)o)) - 1;
                    const uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)cl_g, cur), cn = (uint32_t)__builtin_amdgcn_readlane((int)cl_n, cur),
                                   c_last = (uint32_t)__builtin_amdgcn_readlane((int)cl_last, cur);
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
                    const uint32_t og = (uint32_t)__builtin_amdgcn_readlane((int)cl_g, o), of = (uint32_t)__builtin_amdgcn_readlane((int)cl_first, o),
                                   ol = (uint32_t)__builtin_amdgcn_readlane((int)cl_last, o);
                    if (hg == og && hp >= of && hp <= ol) atomicAdd(&rc.covg[s_cov[h]], 1u);
                }
            }
            for (uint32_t c = i + lane; c < e; c += 64)
                if (s_pos1[c]) fw.cand_pos1[base + c] = handled_mark; // handled
        }
    }
    RC_MARK(7);
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
#undef lane
#undef wave
#undef RC_MARK

// The counters of a launch sequence to their pinned mirror on the host, and zero again on the device for the next batch: one small launch
// where a copy and a memset were two (mapper.cpp launch_lane; to: the device address of the mirror).  Round 6: the superblock counts of the
// filtered sequence (FilterWork::super_count: the filter kernel's waves ADD to them) are cleared here as well, for the next batch.
constexpr int CH_THREADS = 256;
__global__ __launch_bounds__(CH_THREADS) void counters_home_kernel(unsigned long long* from, unsigned long long* to, uint32_t n, uint4* zero, uint32_t n_zero4)
{
    const uint32_t i = threadIdx.x;
    if (i < n) {
        to[i] = from[i];
        from[i] = 0;
    }
    for (uint32_t j = i; j < n_zero4; j += CH_THREADS) zero[j] = make_uint4(0, 0, 0, 0);
}
hipError_t launch_counters_home(unsigned long long* from, unsigned long long* to, uint32_t n, hipStream_t stream, uint32_t* zero, uint32_t n_zero)
{
    if (n > CH_THREADS || (n_zero & 3u) || (reinterpret_cast<uintptr_t>(zero) & 15u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(counters_home_kernel, dim3(1), dim3(CH_THREADS), 0, stream, from, to, n, reinterpret_cast<uint4*>(zero), zero ? n_zero / 4 : 0u);
    return hipGetLastError();
}

// DRPRG_FT_DEBUG=8: no read_cluster_kernel; the generic pipeline runs iff there is a hit
__global__ void flag_complex_kernel(const unsigned long long* n_hits, unsigned long long* n_complex)
{
    if (*n_hits) *n_complex = 1;
}

bool read_cluster_wave_form_requested()
{
#ifdef DRPRG_EXPERIMENTAL
    const char* form = std::getenv("DRPRG_RC_FORM");
    return form && std::string(form) == "wave";
#else
    return false; // (the wave form is part of `make EXPERIMENTAL=1` only)
#endif
}

hipError_t launch_read_cluster(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc, int n_cus, bool skip, hipStream_t stream)
{
    if (skip) {
        hipLaunchKernelGGL(flag_complex_kernel, dim3(1), dim3(1), 0, stream, a.n_hits, rc.n_complex);
        return hipGetLastError();
    }
    ReadClusterArgs rcd = rc;
    // make EXPERIMENTAL=1 + DRPRG_RC_FORM=wave (read per launch; tests switch it): the wave form first (read_cluster_wave.hip: no workgroup barriers;
    // everything a 150-base read needs) and this kernel behind it for what that leaves.  NOT the default: measured on MI355X
    // (profiles/r04/rc_forms.txt) the wave form takes 61 us where this kernel takes 68 us on configs[1], and then still needs this
    // kernel for the reads it left (37 us) and a 8 us kernel for the totals; on configs[4] 2.2 ms against 1.3 ms.  DESIGN.md section 6.
    const bool wave_first = read_cluster_wave_form_requested();
    rcd.second_pass = 0;
    // (long reads do not fit a wave's 128 staged candidates: a batch of them goes straight to the workgroup form)
#ifdef DRPRG_EXPERIMENTAL
    if (wave_first && rc.n_unfit && rc.chunk_flags && a.n_bases / (a.n_reads ? a.n_reads : 1u) <= 600) {
        HIP_TRY(launch_read_cluster_wave(a, fw, rc, n_cus, stream));
        rcd.second_pass = 1;
    }
#else
    (void)wave_first;
#endif
    static unsigned long long* d_phase = nullptr;
    const bool debug = std::getenv("DRPRG_RC_DEBUG") != nullptr;
    if (debug) {
        if (!d_phase) HIP_TRY(hipMalloc(&d_phase, 14 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d_phase, 0, 14 * sizeof(unsigned long long), stream));
        rcd.phase_clock = d_phase;
    }
    size_t dyn = (size_t)rc.n_prgs * sizeof(uint32_t); // the per-PRG histogram, behind ~77 KB of static LDS
    {   // + the shortest paths, 16 bits each, if the same number of workgroups per CU still fits (configs[4]: 500 PRGs, 88 bytes to spare)
        static size_t static_lds = 0;
        if (!static_lds) {
            size_t most = 0;
            for (const void* k : { reinterpret_cast<const void*>(&read_cluster_kernel<false, 128>), reinterpret_cast<const void*>(&read_cluster_kernel<false, 256>),
                     reinterpret_cast<const void*>(&read_cluster_kernel<false, 512>), reinterpret_cast<const void*>(&read_cluster_kernel<true, 128>),
                     reinterpret_cast<const void*>(&read_cluster_kernel<true, 256>), reinterpret_cast<const void*>(&read_cluster_kernel<true, 512>) }) {
                hipFuncAttributes at {};
                HIP_TRY(hipFuncGetAttributes(&at, k));
                most = std::max<size_t>(most, at.sharedSizeBytes);
            }
            static_lds = most;
        }
        const size_t lds_cu = 160 * 1024, with = dyn + (size_t)rc.n_prgs * sizeof(uint16_t);
        const size_t fit_without = std::min<size_t>(lds_cu / (static_lds + dyn), DRPRG_RC_WG_PER_CU), fit_with = std::min<size_t>(lds_cu / (static_lds + with), DRPRG_RC_WG_PER_CU);
        rcd.minpath_in_lds = fit_with == fit_without ? 1u : 0u;
        if (rcd.minpath_in_lds) dyn = with;
    }
    // look-ahead by the batch's mean read length (DRPRG_RC_AHEAD=128 / 256 / 512 forces one; the second pass behind the wave form shares
    // that form's chunk numbering: 512)
    const uint64_t mean_len = a.n_bases / (a.n_reads ? a.n_reads : 1u);
    int ahead = mean_len <= 300 ? 128 : mean_len <= 600 ? 256 : 512;
    if (const char* e = std::getenv("DRPRG_RC_AHEAD")) {
        const int v = std::atoi(e);
        if (v == 128 || v == 256 || v == 512) ahead = v;
    }
    if (rcd.second_pass) ahead = 512;
    static size_t configured[6][MAX_HIP_DEVICES] = {};
    auto launch = [&](auto kernel, size_t (&conf)[MAX_HIP_DEVICES]) -> hipError_t {
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), dyn, conf));
        hipLaunchKernelGGL(kernel, dim3((uint32_t)n_cus * DRPRG_RC_WG_PER_CU), dim3(RC_THREADS), dyn, stream, a, fw, rcd);
        return hipGetLastError();
    };
    if (rcd.slice_prefix) {
        if (ahead == 128) HIP_TRY(launch(&read_cluster_kernel<true, 128>, configured[0]));
        else if (ahead == 256) HIP_TRY(launch(&read_cluster_kernel<true, 256>, configured[1]));
        else HIP_TRY(launch(&read_cluster_kernel<true, 512>, configured[2]));
    } else {
        if (ahead == 128) HIP_TRY(launch(&read_cluster_kernel<false, 128>, configured[3]));
        else if (ahead == 256) HIP_TRY(launch(&read_cluster_kernel<false, 256>, configured[4]));
        else HIP_TRY(launch(&read_cluster_kernel<false, 512>, configured[5]));
    }
    if (debug) { // cycles of thread 0, summed over the workgroups, per phase (the marks follow the barriers of the chunk loop)
        unsigned long long h[14];
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h, d_phase, sizeof h, hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (int i = 0; i < 10; ++i) sum += h[i];
        sum += h[12];
        std::fprintf(stderr, "[read_cluster phases, %% of %llu Mcycles]", sum / 1000000);
        for (int i = 0; i < 10; ++i) std::fprintf(stderr, " %d:%.1f", i, sum ? 100.0 * (double)h[i] / (double)sum : 0.0);
        std::fprintf(stderr, " request:%.1f", sum ? 100.0 * (double)h[12] / (double)sum : 0.0);
        std::fprintf(stderr, " | %llu reads on the wave path in %llu chunks\n", h[10], h[11]);
    }
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
