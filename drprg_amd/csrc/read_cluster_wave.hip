// read_cluster_wave.hip -- the wave form of the last stage of the candidate sequences: clusters, size / overlap filters and coverage
// straight from the ordered candidate list, ONE WAVE PER 64 CANDIDATES AND NO WORKGROUP BARRIER (round 4).
//
// read_cluster.hip does the same work with a 1024-thread workgroup per 1536 candidates and a dozen workgroup barriers per chunk; its
// own phase clocks showed where the time went: sixteen waves waiting at every barrier for the slowest one, a third of a chunk spent on
// the four or five reads with hits in several groups while fifteen waves idle, a quarter (slices form) bisecting for the candidates.
// A short read has at most a few dozen candidates, and everything the algorithm needs of its neighbourhood -- where reads and
// segments start, how many hits a segment has, whether a read's hits lie in one (prg, strand) group -- is a bit mask or a prefix sum
// over the wave: ballots, mbcnt / DPP scans and a few ds_bpermute shuffles, all in registers.
//
// Unit of work: a tile of 96 candidates of the ordered list (ordered by (read, position)) plus the 32 behind it as look-ahead.  A
// read belongs to the tile that holds its first candidate; it is handled here if it ends inside the 128 staged slots, none of
// its positions is beyond 2^16 - 2, and -- hits in several (prg, strand) groups -- it has at most RW_HCAP hits.
// Everything else (long reads above all: a 4 kb Nanopore read has ~250 candidates) is left untouched and counted in rc.n_unfit;
// read_cluster_kernel then runs behind this kernel as a second pass over what is left (it returns at once when nothing is).
//   reads with all hits in one group: the segments (runs of minimizers without a position gap > max_diff) are the clusters and
//     the overlap sweep cannot drop any: a segment is kept iff its hit count exceeds its threshold; its minimizers add their hits.
//   reads with hits in several groups: their hits go to the wave's own 2 KB of LDS and one pass of the cluster algorithm of
//     read_cluster_kernel's wave path runs on them (lane j = cluster j; pandora define_clusters / filter_clusters).
// Semantics are exactly read_cluster_kernel's (DESIGN.md section 4 "Clusters", "Filter"); tests/test_gpu_parity.py holds both forms
// against the oracle.  Opt-in (DRPRG_RC_FORM=wave): it is not faster than the workgroup form, see launch_read_cluster.
#include "filter_common.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>

namespace drprg {
namespace dev {

constexpr int RW_THREADS = 1024;
constexpr int RW_WAVES = RW_THREADS / 64;
constexpr int RW_HCAP = 256;    // hits of a multi-group read that fit the wave's LDS region
constexpr int RW_FAST_REC = 4;  // index records per minimizer whose loads are unrolled (99.4 % of the keys of the mtb-like index have <= 4)
constexpr int RW_OWN = 96;      // candidates a tile owns of its 128 staged ones: a 150-base read has at most ~25 candidates, so 32 slots of
                                // look-ahead do, and every candidate is staged 1.33 times instead of twice

__device__ __forceinline__ uint32_t rw_mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// value of slot `idx` (0..127) of a two-row array: row 0 = slots 0..63, row 1 = slots 64..127 (every lane takes part)
__device__ __forceinline__ uint32_t rw_at(uint32_t v0, uint32_t v1, int idx)
{
    const uint32_t a = (uint32_t)__shfl((int)v0, idx & 63), b = (uint32_t)__shfl((int)v1, idx & 63);
    return idx < 64 ? a : b;
}
// highest set bit at or below slot (row, lane) of the 128-bit mask (m0, m1): -1 if none
__device__ __forceinline__ int rw_prev_set(uint64_t m0, uint64_t m1, int row, int lane_)
{
    const uint64_t le = lane_ == 63 ? ~0ull : ((2ull << lane_) - 1ull);
    if (row == 0) {
        const uint64_t x = m0 & le;
        return x ? 63 - __clzll((long long)x) : -1;
    }
    const uint64_t x = m1 & le;
    if (x) return 127 - __clzll((long long)x);
    return m0 ? 63 - __clzll((long long)m0) : -1;
}
// lowest set bit strictly above slot (row, lane): 128 if none
__device__ __forceinline__ int rw_next_set(uint64_t m0, uint64_t m1, int row, int lane_)
{
    const uint64_t gt = lane_ == 63 ? 0ull : ~((2ull << lane_) - 1ull);
    if (row == 0) {
        const uint64_t x = m0 & gt;
        if (x) return __ffsll((long long)x) - 1;
        return m1 ? 64 + __ffsll((long long)m1) - 1 : 128;
    }
    const uint64_t x = m1 & gt;
    return x ? 64 + __ffsll((long long)x) - 1 : 128;
}
// bits [from, to) of a 128-bit range as (low row, high row) masks
__device__ __forceinline__ void rw_range(int from, int to, uint64_t& r0, uint64_t& r1)
{
    auto below = [](int n) -> uint64_t { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); };
    r0 = below(to) & ~below(from);
    r1 = below(to - 64) & ~below(from - 64);
}

template <bool SLICES>
__global__ __launch_bounds__(RW_THREADS) void read_cluster_wave_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    extern __shared__ uint32_t s_hist[]; // clusters kept per PRG, then the waves' hit regions
    __shared__ uint16_t s_grp_all[RW_WAVES][RW_HCAP];
    __shared__ uint16_t s_hpos_all[RW_WAVES][RW_HCAP];
    __shared__ uint32_t s_cov_all[RW_WAVES][RW_HCAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint16_t* const s_grp = s_grp_all[wave];
    uint16_t* const s_hpos = s_hpos_all[wave];
    uint32_t* const s_cov = s_cov_all[wave];
    if (*reinterpret_cast<volatile uint32_t*>(a.overflow) & 4u) return; // a candidate slice overflowed: the host re-runs the batch
    for (uint32_t i = tid; i < rc.n_prgs; i += RW_THREADS) s_hist[i] = 0;
    __syncthreads(); // (the only barriers of the kernel: this one and the one in front of the histogram's way out)
    const uint32_t total = SLICES ? rc.slice_prefix[rc.n_slices] : *fw.cand_total;
    const uint32_t handled_mark = SLICES ? rc.mark_epoch : 0u;
    const uint32_t dbg = (uint32_t)rc.second_pass >> 8; // DRPRG_RW_DEBUG (timing experiments, wrong results): 1 no coverage atomics, 2 no marks, 4 no record loads
    const uint32_t w1_magic = w1_reciprocal(a.w);
    unsigned long long my_kept_hits = 0;
    uint32_t my_kept = 0, my_unfit = 0;
    const uint32_t n_tiles = (total + RW_OWN - 1) / RW_OWN;
    const uint32_t gwave = blockIdx.x * RW_WAVES + (uint32_t)wave, n_gwaves = gridDim.x * RW_WAVES;
    uint32_t cursor = 0; // SLICES: a slice at or before the one that holds the wave's current tile (tiles are taken in ascending order)

    // dense list: the raw loads of a tile (unconditional: a slot past the end reads the tile's first candidate and drops it)
    struct Staged {
        uint32_t rd[2], p1[2], e_prev, e_next;
        uint4 rec[2];
    };
    auto request = [&](uint32_t t) {
        Staged st;
        const uint32_t b = t * (uint32_t)RW_OWN, n = total - b < 128u ? total - b : 128u;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t i = 64u * (uint32_t)r + (uint32_t)lane;
            const uint32_t d = b + (i < n ? i : 0u);
            st.rd[r] = (uint32_t)fw.cand_info[d];
            st.p1[r] = fw.cand_pos1[d];
            st.rec[r] = fw.cand_rec[d];
        }
        st.e_prev = (uint32_t)fw.cand_info[b ? b - 1 : 0u];
        st.e_next = (uint32_t)fw.cand_info[b + 128u < total ? b + 128u : 0u];
        return st;
    };
    Staged nx {};
    if constexpr (!SLICES)
        if (gwave < n_tiles) nx = request(gwave);
    (void)request;

    for (uint32_t tile = gwave; tile < n_tiles; tile += n_gwaves) {
        const uint32_t base = tile * (uint32_t)RW_OWN;
        const uint32_t n_in = total - base < 128u ? total - base : 128u;
        // ---- the 128 slots: read, position + 1, record; the candidate before the tile and the one behind the staged range ----
        uint32_t rd[2], p1[2];
        uint4 rec[2];
        size_t src[2] = { 0, 0 }; // SLICES: where the slot's candidate lives in the slice arrays
        (void)src;
        uint32_t prev_read = 0xFFFFFFFFu, next_read = 0xFFFFFFFFu;
        if constexpr (!SLICES) {
            // (requested one tile ago: the candidates of this tile have been travelling while the last one was worked on)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                rd[r] = nx.rd[r];
                p1[r] = nx.p1[r];
                rec[r] = nx.rec[r];
            }
            if (base) prev_read = nx.e_prev & 0x7FFFFFFFu;
            if (base + 128u < total) next_read = nx.e_next & 0x7FFFFFFFu;
            nx = request(tile + n_gwaves < n_tiles ? tile + n_gwaves : tile);
        } else {
            // entry d of the ordered list lives in the slice s with P[s] <= d < P[s + 1].  The wave keeps a cursor; 64 prefix entries from
            // the cursor on go to the lanes and every slot bisects them with shuffles.  Slices so sparsely filled that 130 entries span
            // more than 63 of them: every slot bisects the prefix in memory instead.
            const uint32_t* __restrict__ P = rc.slice_prefix;
            const uint32_t d_first = base ? base - 1 : 0u;
            { // advance the cursor to the slice of entry d_first (a gallop from where the last tile left it)
                uint32_t lo = cursor, step = 64;
                while (lo + step < rc.n_slices && P[lo + step] <= d_first) {
                    lo += step;
                    step <<= 1;
                }
                uint32_t hi = lo + step < rc.n_slices ? lo + step : rc.n_slices; // P[lo] <= d_first < P[hi] (or hi == n_slices)
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (P[mid] <= d_first) lo = mid;
                    else hi = mid;
                }
                cursor = lo;
            }
            const uint32_t pl = cursor + (uint32_t)lane <= rc.n_slices ? P[cursor + (uint32_t)lane] : 0xFFFFFFFFu; // ascending over the lanes
            const uint32_t d_last = base + n_in; // (the entry behind the staged range, if there is one)
            const bool in_lanes = (uint32_t)__shfl((int)pl, 63) > d_last || cursor + 63u >= rc.n_slices;
            auto locate = [&](uint32_t d) -> size_t {
                if (in_lanes) { // the largest lane j with pl[j] <= d
                    int lo = 0;
#pragma unroll
                    for (int stepb = 32; stepb > 0; stepb >>= 1) {
                        const uint32_t v = (uint32_t)__shfl((int)pl, lo + stepb);
                        if (v <= d) lo += stepb;
                    }
                    const uint32_t pv = (uint32_t)__shfl((int)pl, lo);
                    return (size_t)(cursor + (uint32_t)lo) * a.tile_cap + (d - pv);
                }
                uint32_t lo = cursor, hi = rc.n_slices;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (P[mid] <= d) lo = mid;
                    else hi = mid;
                }
                return (size_t)lo * a.tile_cap + (d - P[lo]);
            };
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t i = 64u * (uint32_t)r + (uint32_t)lane;
                const uint32_t d = base + (i < n_in ? i : 0u);
                src[r] = locate(d);
                rd[r] = (uint32_t)a.tile_info[src[r]];
                p1[r] = a.tile_pos1[src[r]];
                rec[r] = a.tile_rec[src[r]];
            }
            // (every lane takes part in the shuffles of locate: uniform branches only)
            const size_t sp = locate(d_first), sn = locate(base + 128u < total ? base + 128u : d_first);
            const uint32_t e_prev = (uint32_t)a.tile_info[sp], e_next = (uint32_t)a.tile_info[sn];
            if (base) prev_read = e_prev & 0x7FFFFFFFu;
            if (base + 128u < total) next_read = e_next & 0x7FFFFFFFu;
        }
        bool valid[2];
        uint32_t cnt[2], grp[2], pos[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t i = 64u * (uint32_t)r + (uint32_t)lane;
            valid[r] = i < n_in;
            rd[r] = valid[r] ? rd[r] & 0x7FFFFFFFu : READ_NONE;
            if (!valid[r]) p1[r] = 0;
            if constexpr (SLICES) { // (handled already: a batch run again)
                const uint32_t mk = fw.cand_pos1[base + (i < n_in ? i : 0u)];
                if (mk == handled_mark) p1[r] = 0;
            }
            cnt[r] = p1[r] ? rec[r].y : 0u;
            grp[r] = (rec[r].z >> 16) & 0x7FFFu;
            pos[r] = p1[r] - 1u;
        }
        // ---- read starts, minimizers ----
        const uint32_t rd_up0 = (uint32_t)__shfl_up((int)rd[0], 1), rd_up1 = (uint32_t)__shfl_up((int)rd[1], 1);
        const uint32_t rd_l63 = (uint32_t)__builtin_amdgcn_readlane((int)rd[0], 63);
        const bool rs0 = valid[0] && rd[0] != (lane ? rd_up0 : prev_read), rs1 = valid[1] && rd[1] != (lane ? rd_up1 : rd_l63);
        const uint64_t RS0 = __ballot(rs0), RS1 = __ballot(rs1);
        const uint64_t M0 = __ballot(cnt[0] != 0 || p1[0] != 0), M1 = __ballot(cnt[1] != 0 || p1[1] != 0);
        if (!(M0 | M1)) continue; // nothing but candidates that are no minimizers (or handled already)
        // ---- every minimizer: the minimizer before it (same read?  position gap?  another group?) ----
        const uint64_t lt = (1ull << lane) - 1ull;
        bool seg_start[2], irr[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint64_t m0 = r == 0 ? M0 & lt : M0, m1 = r == 0 ? 0ull : M1 & lt;
            const int pi = m1 ? 127 - __clzll((long long)m1) : (m0 ? 63 - __clzll((long long)m0) : -1); // slot of the previous minimizer
            const int q = pi < 0 ? 0 : pi;
            const uint32_t prd = rw_at(rd[0], rd[1], q), pp1 = rw_at(p1[0], p1[1], q), pg = rw_at(grp[0], grp[1], q);
            const bool is_min = p1[r] != 0;
            const bool same = pi >= 0 && prd == rd[r];
            seg_start[r] = is_min && (!same || (int)(p1[r] - pp1) > rc.max_diff);
            irr[r] = is_min && same && pg != grp[r];
        }
        // ---- the further index records of a minimizer (a k-mer that several k-mer nodes share): another group makes the read irregular;
        // a position beyond 2^16 - 2 makes it unfit for the LDS path ----
        // (the loads of both rows are requested before the first is looked at, without a branch of their own around any of them: one after
        // the other -- a loop per lane over its records -- every record was a round trip to the L2 and the wave as slow as its longest
        // list.  Records 1 .. RW_FAST_REC - 1 this way, and only as far as some lane of the wave has that many; the rare minimizer with
        // more takes the loop after all.)
        bool unfit[2] = { false, false };
        uint32_t xcov[2][RW_FAST_REC - 1]; // coverage index of records 1 .. RW_FAST_REC - 1
        {
            uint32_t kn[2][RW_FAST_REC - 1], pg[2][RW_FAST_REC - 1];
            const uint32_t cmax = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
            const bool any2 = __ballot(cmax > 1) != 0, any3 = __ballot(cmax > 2) != 0, any4 = __ballot(cmax > 3) != 0, any5 = __ballot(cmax > RW_FAST_REC) != 0;
            const bool anyq[3] = { any2, any3, any4 };
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unfit[r] = p1[r] && pos[r] >= 0xFFFEu; // (positions travel as 16 bits in the LDS path)
#pragma unroll
                for (int q = 1; q < RW_FAST_REC; ++q) {
                    kn[r][q - 1] = pg[r][q - 1] = 0;
                    if (anyq[q - 1] && !(dbg & 4u)) { // wave-uniform
                        const uint32_t ri = (uint32_t)q < cnt[r] ? rec[r].x + (uint32_t)q : 0u; // (a lane without that record reads record 0 and ignores it)
                        kn[r][q - 1] = a.rec_knode[ri];
                        pg[r][q - 1] = (uint32_t)a.rec_prg[ri];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t strand = rec[r].z >> 31;
#pragma unroll
                for (int q = 1; q < RW_FAST_REC; ++q) {
                    const bool on = (uint32_t)q < cnt[r];
                    const uint32_t rev = ((kn[r][q - 1] & 1u) == strand) ? 0u : 1u;
                    irr[r] |= on && ((pg[r][q - 1] << 1) | rev) != grp[r];
                    xcov[r][q - 1] = (kn[r][q - 1] >> 1) * 2u + rev;
                }
                if (any5) // (wave-uniform) some minimizer of the tile has more records than that: its lane walks the rest
                    for (uint32_t q = RW_FAST_REC; q < cnt[r]; ++q) {
                        const uint32_t knq = a.rec_knode[rec[r].x + q], pgq = a.rec_prg[rec[r].x + q];
                        const uint32_t rev = ((knq & 1u) == strand) ? 0u : 1u;
                        irr[r] |= ((pgq << 1) | rev) != grp[r];
                    }
            }
        }
        const uint64_t S0 = __ballot(seg_start[0]), S1 = __ballot(seg_start[1]);
        const uint64_t I0 = __ballot(irr[0]), I1 = __ballot(irr[1]);
        const uint64_t U0 = __ballot(unfit[0]), U1 = __ballot(unfit[1]);
        // ---- hit prefix sums over the 128 slots ----
        uint32_t incl[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            uint32_t v = cnt[r];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t n = (uint32_t)__shfl_up((int)v, off);
                if (lane >= off) v += n;
            }
            incl[r] = v;
        }
        incl[1] += (uint32_t)__builtin_amdgcn_readlane((int)incl[0], 63);
        // ---- per slot: its read [F, E), is the read mine, does it fit, is it irregular; per segment start: the segment's hits ----
        uint32_t seg_total[2], read_hits[2];
        int Fs[2], Es[2];
        bool mine[2], fit[2], irregular[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int F = rw_prev_set(RS0, RS1, r, lane);       // first slot of this slot's read (-1: the read began before the tile)
            const int E = rw_next_set(RS0, RS1, r, lane);       // first slot of the next read (128: none staged)
            Fs[r] = F;
            Es[r] = E;
            mine[r] = F >= 0 && F < RW_OWN; // (a read that began before the tile is its own tile's)
            const uint32_t last_rd = (uint32_t)__builtin_amdgcn_readlane((int)rd[1], 63);
            const bool runs_on = E == 128 && n_in == 128u && next_read == last_rd && rd[r] == last_rd; // continues behind the staged range
            uint64_t r0m, r1m;
            rw_range(F < 0 ? 0 : F, E, r0m, r1m);
            fit[r] = !runs_on && !((U0 & r0m) | (U1 & r1m));
            irregular[r] = ((I0 & r0m) | (I1 & r1m)) != 0;
            const int e = rw_next_set(S0, S1, r, lane); // end of the segment that starts here: the next segment start
            const int e1 = (e < E ? e : E) - 1;         // last slot of the segment (a read's first minimizer is a segment start, so e <= E holds
                                                        // whenever the next read has a minimizer; reads without one end the segment too)
            const uint32_t upto = rw_at(incl[0], incl[1], e1 < 0 ? 0 : e1);
            seg_total[r] = upto - (incl[r] - cnt[r]);
            const uint32_t r_end = rw_at(incl[0], incl[1], E - 1), r_beg_incl = rw_at(incl[0], incl[1], F < 0 ? 0 : F),
                           r_beg_cnt = rw_at(cnt[0], cnt[1], F < 0 ? 0 : F);
            read_hits[r] = r_end - (r_beg_incl - r_beg_cnt);
        }
        // ---- decisions at the segment starts of my reads ----
        uint32_t dec[2]; // 0 leave alone, 1 handled, 2 handled and every hit counts, 3 irregular: the wave path below
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            dec[r] = 0;
            if (!(seg_start[r] && mine[r] && fit[r])) continue;
            if (irregular[r]) {
                // (all hits of the read together do not exceed the smallest threshold a cluster can have: nothing can be kept)
                dec[r] = read_hits[r] <= rc.min_cluster_size ? 1u : (read_hits[r] <= (uint32_t)RW_HCAP ? 3u : 0u); // (the same for every segment of the read)
                continue;
            }
            dec[r] = 1;
            if (seg_total[r] > (rec[r].z & 0xFFFFu)) {
                dec[r] = 2;
                atomicAdd(&s_hist[grp[r] >> 1], 1u);
                ++my_kept;
                my_kept_hits += seg_total[r];
            }
        }
        // ---- every minimizer takes the decision of its segment; kept ones add their hits, handled ones are marked ----
        uint32_t mdec[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int sgi = rw_prev_set(S0, S1, r, lane);
            const uint32_t d = rw_at(dec[0], dec[1], sgi < 0 ? 0 : sgi);
            mdec[r] = (p1[r] != 0 && sgi >= 0 && mine[r]) ? d : 0u;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if ((mdec[r] == 1 || mdec[r] == 2) && !(dbg & 2u)) fw.cand_pos1[base + 64u * (uint32_t)r + (uint32_t)lane] = handled_mark;
            if (mdec[r] == 2 && !(dbg & 1u)) {
                atomicAdd(&rc.covg[rec[r].w], 1u);
#pragma unroll
                for (int q = 1; q < RW_FAST_REC; ++q)
                    if ((uint32_t)q < cnt[r]) atomicAdd(&rc.covg[xcov[r][q - 1]], 1u);
                const uint32_t strand = rec[r].z >> 31;
                for (uint32_t q = RW_FAST_REC; q < cnt[r]; ++q) { // (rare)
                    const uint32_t knq = a.rec_knode[rec[r].x + q];
                    atomicAdd(&rc.covg[(knq >> 1) * 2u + (((knq & 1u) == strand) ? 0u : 1u)], 1u);
                }
            }
        }
        // ---- reads that are mine, hold a minimizer (or run on behind the staged range) and do not fit: left to the pass behind this kernel ----
        bool rstart[2] = { rs0, rs1 };
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            uint64_t r0m, r1m;
            rw_range(Fs[r] < 0 ? 0 : Fs[r], Es[r], r0m, r1m);
            const bool has_min = ((M0 & r0m) | (M1 & r1m)) != 0;
            const bool uf = rstart[r] && mine[r] && !fit[r] && (has_min || Es[r] == 128);
            if (uf) rc.chunk_flags[(base + 64u * (uint32_t)r + (uint32_t)lane) / RC_CHUNK_OWN] = 1u;
            const uint64_t UF = __ballot(uf);
            if (lane == 0) my_unfit += (uint32_t)__popcll(UF);
        }
        // ---- irregular reads, one after the other: the read's hits to the wave's LDS, then clusters per group split at gaps, size
        // threshold, the overlap sweep (read_cluster_kernel's wave path, on this wave) ----
        uint64_t todo0 = __ballot(rs0 && mine[0] && fit[0] && irregular[0]), todo1 = __ballot(rs1 && mine[1] && fit[1] && irregular[1]); // read starts
        while (todo0 | todo1) {
            int f;
            if (todo0) {
                f = __ffsll((long long)todo0) - 1;
                todo0 &= todo0 - 1;
            } else {
                f = 64 + __ffsll((long long)todo1) - 1;
                todo1 &= todo1 - 1;
            }
            const int E = (int)rw_at((uint32_t)Es[0], (uint32_t)Es[1], f);
            uint64_t r0m, r1m;
            rw_range(f, E, r0m, r1m);
            // the decision of the read's first segment start (all its segments agree by now: 3, 1 or 0)
            const uint64_t fs0 = S0 & r0m, fs1 = S1 & r1m;
            if (!(fs0 | fs1)) continue; // no minimizer at all
            const int first_seg = fs0 ? __ffsll((long long)fs0) - 1 : 64 + __ffsll((long long)fs1) - 1;
            const uint32_t rdec = rw_at(dec[0], dec[1], first_seg);
            if (rdec == 0) {
                if (lane == 0) {
                    ++my_unfit;
                    rc.chunk_flags[(base + (uint32_t)f) / RC_CHUNK_OWN] = 1u;
                }
                continue;
            }
            if (rdec != 3) continue; // dropped as a whole: marked above
            const uint32_t h_base = rw_at(incl[0], incl[1], f) - rw_at(cnt[0], cnt[1], f); // hits before the read
            const uint32_t n_hits = rw_at(read_hits[0], read_hits[1], f);
            const uint32_t read = rw_at(rd[0], rd[1], f);
            const uint64_t len = a.offsets[read + 1] - a.offsets[read];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool in = ((r == 0 ? r0m : r1m) >> lane) & 1ull;
                if (!in || !cnt[r]) continue;
                const uint32_t h0 = incl[r] - cnt[r] - h_base, strand = rec[r].z >> 31;
                s_grp[h0] = (uint16_t)grp[r];
                s_hpos[h0] = (uint16_t)pos[r];
                s_cov[h0] = rec[r].w;
                for (uint32_t q = 1; q < cnt[r]; ++q) {
                    const uint32_t kn = a.rec_knode[rec[r].x + q], prg = a.rec_prg[rec[r].x + q];
                    const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
                    s_grp[h0 + q] = (uint16_t)((prg << 1) | rev);
                    s_hpos[h0 + q] = (uint16_t)pos[r];
                    s_cov[h0 + q] = (kn >> 1) * 2u + rev;
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // -- clusters (lane j holds cluster j) --
            uint32_t cl_g = 0, cl_n = 0, cl_first = 0, cl_last = 0;
            int nc = 0;
            bool complex = false;
            for (uint32_t b = 0; b < n_hits && !complex; b += 64) {
                const uint32_t h = b + (uint32_t)lane;
                const bool hv = h < n_hits;
                const uint32_t hg = hv ? s_grp[h] : 0u, hp = hv ? s_hpos[h] : 0u;
                uint64_t todo = __ballot(hv);
                while (todo) {
                    const uint32_t G = (uint32_t)__builtin_amdgcn_readlane((int)hg, __ffsll((long long)todo) - 1);
                    const uint64_t m = __ballot(hv && hg == G);
                    todo &= ~m;
                    const uint64_t open = __ballot(lane < nc && cl_g == G); // the group's clusters so far: the highest lane is the latest
                    const int fo = open ? 63 - __clzll((long long)open) : -1;
                    const uint32_t open_last = fo >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)cl_last, fo) : 0u;
                    const uint64_t below = m & lt;
                    const int prev = below ? 63 - __clzll((long long)below) : -1;
                    const uint32_t prev_pos = (uint32_t)__shfl((int)hp, prev < 0 ? lane : prev);
                    const bool member = ((m >> lane) & 1ull) != 0;
                    const bool start = member && (prev >= 0 ? (int)(hp - prev_pos) > rc.max_diff : (fo < 0 || (int)(hp - open_last) > rc.max_diff));
                    const uint64_t starts = __ballot(start);
                    const int n_new = __popcll(starts);
                    if (nc + n_new > 64) {
                        complex = true;
                        break;
                    }
                    const uint64_t cont = m & (starts ? (1ull << (__ffsll((long long)starts) - 1)) - 1ull : ~0ull); // before the first start
                    if (cont) {
                        const uint32_t lp = (uint32_t)__builtin_amdgcn_readlane((int)hp, 63 - __clzll((long long)cont));
                        if (lane == fo) {
                            cl_n += (uint32_t)__popcll(cont);
                            cl_last = lp;
                        }
                    }
                    const int c = lane - nc; // lane nc + c takes the c-th start
                    const bool take = c >= 0 && c < n_new;
                    int first_lane = lane, last_lane = lane;
                    uint32_t span_n = 0;
                    if (take) {
                        uint64_t mm = starts;
                        for (int k = 0; k < c; ++k) mm &= mm - 1;
                        first_lane = __ffsll((long long)mm) - 1;
                        const uint64_t rest = mm & (mm - 1);
                        const uint64_t upto = rest ? (1ull << (__ffsll((long long)rest) - 1)) - 1ull : ~0ull;
                        const uint64_t span = m & upto & ~((1ull << first_lane) - 1ull);
                        last_lane = 63 - __clzll((long long)span);
                        span_n = (uint32_t)__popcll(span);
                    }
                    const uint32_t fpos = (uint32_t)__shfl((int)hp, first_lane), lpos = (uint32_t)__shfl((int)hp, last_lane);
                    if (take) {
                        cl_g = G;
                        cl_n = span_n;
                        cl_first = fpos;
                        cl_last = lpos;
                    }
                    nc += n_new;
                }
            }
            if (complex) { // more than 64 clusters: left alone
                if (lane == 0) {
                    ++my_unfit;
                    rc.chunk_flags[(base + (uint32_t)f) / RC_CHUNK_OWN] = 1u;
                }
                continue;
            }
            const uint64_t expected = expected_minimizers(len, a.w, w1_magic);
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
                    const uint32_t of = (uint32_t)__builtin_amdgcn_readlane((int)cl_first, o), on = (uint32_t)__builtin_amdgcn_readlane((int)cl_n, o),
                                   og = (uint32_t)__builtin_amdgcn_readlane((int)cl_g, o);
                    rank += (of < cl_first || (of == cl_first && (on > cl_n || (on == cl_n && og < cl_g)))) ? 1u : 0u;
                }
                const int nk = __popcll(alive);
                int prevc = -1;
                uint32_t pg = 0, pn = 0, p_last = 0;
                for (int o = 0; o < nk; ++o) {
                    const int cur = __ffsll((long long)__ballot(kept && rank == (uint32_t)o)) - 1;
                    const uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)cl_g, cur), cn = (uint32_t)__builtin_amdgcn_readlane((int)cl_n, cur),
                                   c_last = (uint32_t)__builtin_amdgcn_readlane((int)cl_last, cur);
                    if (prevc >= 0) {
                        const bool same_prg_other_strand = (pg >> 1) == (cg >> 1) && (pg & 1u) != (cg & 1u);
                        if (same_prg_other_strand || c_last <= p_last) {
                            if (pn >= cn) {
                                alive &= ~(1ull << cur);
                                continue;
                            }
                            alive &= ~(1ull << prevc);
                        }
                    }
                    prevc = cur;
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
            for (uint32_t b = 0; b < n_hits; b += 64) { // every hit finds its cluster among the survivors
                const uint32_t h = b + (uint32_t)lane;
                const uint32_t hg = h < n_hits ? s_grp[h] : 0xFFFFFFFFu, hp = h < n_hits ? s_hpos[h] : 0u;
                for (uint64_t mm = alive; mm; mm &= mm - 1) {
                    const int o = __ffsll((long long)mm) - 1;
                    const uint32_t og = (uint32_t)__builtin_amdgcn_readlane((int)cl_g, o), of = (uint32_t)__builtin_amdgcn_readlane((int)cl_first, o),
                                   ol = (uint32_t)__builtin_amdgcn_readlane((int)cl_last, o);
                    if (hg == og && hp >= of && hp <= ol) atomicAdd(&rc.covg[s_cov[h]], 1u);
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool in = ((r == 0 ? r0m : r1m) >> lane) & 1ull;
                if (in && p1[r]) fw.cand_pos1[base + 64u * (uint32_t)r + (uint32_t)lane] = handled_mark;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- totals.  Every workgroup's histogram and counters go to a slot of its own in global memory -- plain stores --, and a
    // one-workgroup kernel behind this one adds the slots up (rw_totals_kernel).  What was tried first: one atomic per wave on the
    // three batch counters -- 8192 waves finishing together on one cache line, ~10 ns each in the L2: 190 us of a 230 us kernel --, and
    // the last workgroup summing the slots behind a __threadfence(): the fence is an L2 write-back on this chip, 512 of them: 130 us. ----
    __shared__ uint32_t s_tot[4];
    if (tid < 4) s_tot[tid] = 0;
    __syncthreads();
    {
        unsigned long long kh = my_kept_hits;
        uint32_t kp = my_kept, uf = my_unfit;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kh += (unsigned long long)__shfl_xor((long long)kh, off);
            kp += (uint32_t)__shfl_xor((int)kp, off);
            uf += (uint32_t)__shfl_xor((int)uf, off);
        }
        if (lane == 0) {
            if (kp) atomicAdd(&s_tot[0], kp);
            if (kh) atomicAdd(&s_tot[1], (uint32_t)kh); // (a workgroup's share of a batch stays far below 2^32 hits)
            if (uf) atomicAdd(&s_tot[2], uf);
        }
    }
    __syncthreads();
    const uint32_t n_slot = rc.n_prgs + 4;
    uint32_t* const mine_slot = rc.wg_partials + (size_t)blockIdx.x * n_slot;
    for (uint32_t i = tid; i < n_slot; i += RW_THREADS) mine_slot[i] = i < rc.n_prgs ? s_hist[i] : s_tot[i - rc.n_prgs];
}

// the slots of read_cluster_wave_kernel's workgroups summed into the batch's vectors and counters (one workgroup)
__global__ __launch_bounds__(RW_THREADS) void rw_totals_kernel(SketchArgs a, ReadClusterArgs rc, uint32_t n_wg)
{
    if (*reinterpret_cast<volatile uint32_t*>(a.overflow) & 4u) return; // (the slots were not written)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_slot = rc.n_prgs + 4;
    extern __shared__ uint32_t s_hist[];
    __shared__ unsigned long long s_sum[4];
    for (uint32_t i = tid; i < rc.n_prgs; i += RW_THREADS) s_hist[i] = 0;
    if (tid < 4) s_sum[tid] = 0;
    __syncthreads();
    auto ld = [&](uint32_t g, uint32_t i) -> uint32_t { return g < n_wg && i < n_slot ? rc.wg_partials[(size_t)g * n_slot + i] : 0u; };
    auto acc = [&](uint32_t i, uint32_t v) {
        if (!v) return;
        if (i < rc.n_prgs) atomicAdd(&s_hist[i], v);
        else atomicAdd(&s_sum[i - rc.n_prgs], (unsigned long long)v);
    };
    // every wave takes every sixteenth group of four slot rows, the four rows' loads in flight together
    for (uint32_t g0 = (uint32_t)wave * 4u; g0 < n_wg; g0 += RW_WAVES * 4u)
        for (uint32_t i = (uint32_t)lane; i < n_slot; i += 64u) {
            const uint32_t v0 = ld(g0, i), v1 = ld(g0 + 1, i), v2 = ld(g0 + 2, i), v3 = ld(g0 + 3, i);
            acc(i, v0);
            acc(i, v1);
            acc(i, v2);
            acc(i, v3);
        }
    __syncthreads();
    for (uint32_t i = tid; i < rc.n_prgs; i += RW_THREADS)
        if (s_hist[i]) atomicAdd(&rc.prg_reads[i], s_hist[i]);
    if (tid == 0 && s_sum[0]) atomicAdd(rc.n_clusters_kept, s_sum[0]);
    if (tid == 1 && s_sum[1]) atomicAdd(rc.n_hits_kept, s_sum[1]);
    if (tid == 2 && s_sum[2]) atomicAdd(rc.n_unfit, s_sum[2]);
}

hipError_t launch_read_cluster_wave(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc_in, int n_cus, hipStream_t stream)
{
    ReadClusterArgs rc = rc_in;
    if (const char* e = std::getenv("DRPRG_RW_DEBUG")) rc.second_pass |= std::atoi(e) << 8;
    const size_t dyn = (size_t)rc.n_prgs * sizeof(uint32_t);
    const uint32_t grid = std::min<uint32_t>((uint32_t)n_cus * 2, RC_WAVE_MAX_WG); // two workgroups of 16 waves per CU
    if (rc.slice_prefix) {
        static size_t configured_slices[MAX_HIP_DEVICES] = {};
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&read_cluster_wave_kernel<true>), dyn, configured_slices));
        hipLaunchKernelGGL(read_cluster_wave_kernel<true>, dim3(grid), dim3(RW_THREADS), dyn, stream, a, fw, rc);
    } else {
        static size_t configured[MAX_HIP_DEVICES] = {};
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&read_cluster_wave_kernel<false>), dyn, configured));
        hipLaunchKernelGGL(read_cluster_wave_kernel<false>, dim3(grid), dim3(RW_THREADS), dyn, stream, a, fw, rc);
    }
    HIP_TRY(hipGetLastError());
    static size_t configured_totals[MAX_HIP_DEVICES] = {};
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&rw_totals_kernel), dyn, configured_totals));
    hipLaunchKernelGGL(rw_totals_kernel, dim3(1), dim3(RW_THREADS), dyn, stream, a, rc, grid);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
