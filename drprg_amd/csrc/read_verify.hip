// read_verify.hip -- the verification stage of the filtered launch sequence for short-read batches, READ BY READ (round 5).
// Opt-in (DRPRG_VERIFY_FORM=read): bit-exact, and not faster than verify_count_kernel -- measurements at the end of this comment.
//
// Where it sits (overview at the top of sketch_filter.hip): sketch_filter_kernel leaves the positions of the read k-mers that may be
// index k-mers, cand_scan / cand_gather make one dense list of them ordered by position (fw.cand_gp), and a verification kernel
// turns every entry into the candidate record read_cluster_kernel consumes (cand_info / cand_pos1 / cand_rec): exact index lookup +
// "is this k-mer a (w,k) window minimizer of its read".  Inside the external `pandora map` process that
// /root/reference/src/lib.rs:580-642 spawns this is Seq::minimizer_sketch + the index probe of add_read_hits.
//
// verify_count_kernel (candidates.hip) answers the minimizer question per CANDIDATE: one lane hashes the 2w-1 neighbouring k-mers
// of its candidate.  A read that comes from the panel carries ~25 candidates ~6 positions apart, so its 136 k-mers are hashed
// 25 x 21 / 136 = 3.9 times over (54 M VALU wave-instructions per 10 M reads, 0.116 ms; 0.82 ms on the 8-fold
// index).  This kernel answers it per READ.  Every WAVE works on its own (no workgroup barrier before the totals): it takes a chunk
// of 192 consecutive candidates (three per lane) plus 64 of look-ahead, finds their reads (the candidates of a read sit next to
// each other: ballots over "my predecessor lies in another read" give the runs), and every read that holds at least RV_DENSE_MIN
// candidates is sketched ONCE -- its bases laid out 16 per lane over as many lanes as it needs, reads back to back, hashed and
// window-minimised in registers by the block sketch_wave_kernel is made of (sketch_block.h; neighbours by DPP wave shifts) -- after
// which each candidate only looks its own position up (minimizer bit, canonical hash, strand: the wave's LDS), probes the exact
// table if it is a minimizer and writes its record.  Reads with fewer candidates ride along in the lanes the last pass of a chunk
// has free; the candidates of the others (an off-panel read that shares a 15-mer with the index, or just passed the Bloom filter:
// 28 % of all candidates on the 8d index) wait in a queue in LDS and go through the lane path 64 at a time, in two steps with a
// second queue between them: probe_one_lane (the lookup alone: most of them end there) and verify_one_lane (verify_lane.h).
// Same outputs as verify_count_kernel, bit for bit where it matters downstream (cand_pos1, the read of every candidate, slot /
// strand / record of every minimizer, the per-workgroup totals).
//
// Ownership.  Chunk c = candidates [c OWN, c OWN + OWN + 64) of the list: the last 64 slots are look-ahead.  A read belongs to
// the chunk whose OWNED range holds its first candidate; that chunk handles the read's candidates up to the end of its staged
// range, the ones beyond ("overhang": a read with more candidates than that) are handled one by one, through the queue, by the
// chunk that owns their list index.  Every decision is taken from positions alone (fw.cand_gp is never written here), so every
// candidate is written by exactly one wave.
//
// The sketch of a chunk.  The sketched reads of a chunk are laid out as one stream of 16-base pieces (piece i of a read = its bases
// [A + 16 i, A + 16 i + 16), A = the read's start rounded down to 16, so that every piece is one aligned 16-byte load -- or one word
// of a 2-bit packed batch).  Pass p evaluates pieces [61 p, 61 p + 61) in lanes 1..61; lane 0 and lanes 62, 63 hold the pieces
// before and after and only supply neighbours (the same tiling as sketch_wave_kernel), so a read may run from one pass into the
// next.  Reads never see each other's k-mers: a k-mer is valid iff it starts at or after its read's first base and ends inside
// the read, an invalid k-mer is 0, a window holding one has minimum 0, and the last K-1 >= 1 positions of every read are invalid.
//
// Measured (MI355X, 10 M x 150 bp, profiles/r05/read_verify.txt; rocprofv3 averages): 122 us against verify_count_kernel's 113 on
// the 8d index, 231 against 218 on the 2-fold, 870 against 838 on the 8-fold.  Why the saved hashes do not show: both kernels run at
// ~5 cycles per wave-instruction per SIMD, scalar instructions included, and this one executes 33.7 M VALU + 10.3 M SALU + 1.6 M
// LDS / VMEM per batch where the lane form executes 54 M + ~3 M.  Of the 33.7 M: 17.9 M are the sketch itself (17,908 passes of
// ~1000 for 1.04 M pieces -- 95 % full lanes), ~2.3 M the look-ups behind every pass, 6.9 M the bookkeeping of 9,400 chunks (positions,
// reads, runs, three prefix sums, states), 6.5 M the lane path for the 499 k candidates of reads that are not sketched (14.7 M before
// it got its lookup-only first step).  A first form with one workgroup per chunk and barriers between the phases took 199 us (a chain
// of dependent memory round trips per chunk, five chunk streams per CU); hoisted per-lane invariants and SGPR pressure cost up to
// 220 bytes of scratch per lane until the lane index was made opaque to the compiler per chunk.  What would have to go for a win: the
// per-chunk bookkeeping and the scalar instructions (a quarter of the kernel), not the hashes.
#include "filter_common.h"
#include "sketch_block.h"
#include "verify_lane.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>

namespace drprg {
namespace dev {

constexpr int RV_THREADS = 256;
constexpr int RV_WAVES = RV_THREADS / 64;     // four autonomous waves per workgroup: they only meet for the workgroup's totals
#ifndef DRPRG_RV_ROUNDS // (build-time knobs of measurement builds: Makefile EXTRA_DEFS)
#define DRPRG_RV_ROUNDS 3
#define DRPRG_RV_WAVES_PER_SIMD 4
#endif
constexpr int RV_ROUNDS = DRPRG_RV_ROUNDS;    // owned slots per lane of a full chunk: slot s = 64 r + lane
constexpr int RV_OWN = 64 * RV_ROUNDS;        // candidates a full chunk owns; one more round of 64 is look-ahead
constexpr int RV_EVAL = 61;                   // pieces a pass evaluates (lanes 1..61)
constexpr int RV_DENSE_MIN = 5;               // candidates a read must hold in the chunk to be sketched: a sketch costs a read of 150
                                              // bases ~11 lanes x 900 / 61 wave-instructions, the lane path ~30 per candidate
constexpr int RV_MAX_LEN = 1008;              // longer reads (stray ones in a short-read batch) take the lane path (64 pieces at most)
constexpr int RV_MAXPIECES = 1023;            // pieces per chunk at most (the state word of a candidate)
constexpr int RV_MAXLEAD = 40;                // sketched reads per chunk at most (192 / RV_DENSE_MIN = 38)
constexpr int RV_QCAP = 64 + 4 * 64;          // queue of candidates for the lane path: < 64 waiting + a chunk's slots
constexpr uint32_t RV_NONE = 0xFFFFFFFFu;

// LDS of one wave (nothing in it is shared between waves)
struct alignas(16) RvWave {
    uint4 hv4[RV_EVAL * 4];        // the pass at hand, per evaluated piece: canonical hash + 1 of its 16 positions (0 = invalid)
    uint32_t bits[64];             // ... and minimizer bits | strand bits << 16
    uint64_t ld_r0[RV_MAXLEAD];    // per sketched read: first base,
    uint32_t ld_len[RV_MAXLEAD];   // length,
    uint32_t ld_read[RV_MAXLEAD];  // read,
    uint32_t ld_pc0[64];           // first piece (ascending; entries from the number of sketched reads on: the number of pieces)
    uint32_t s_lead[RV_OWN];       // at a read's first slot: first piece | index among the sketched reads << 16 (RV_NONE: the lane path)
    uint32_t q[RV_QCAP];           // the lane path's queue: list indices of the candidates of reads that are not sketched ...
    uint32_t q2[128];              // ... and of those among them that turn out to be index k-mers: they get the window scan
};

__device__ __forceinline__ void rv_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t rv_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t rv_readlane64(uint64_t v, int l)
{
    return (uint64_t)rv_readlane((uint32_t)v, l) | ((uint64_t)rv_readlane((uint32_t)(v >> 32), l) << 32);
}

// DRPRG_FT_DEBUG=1024: what the kernel did, summed over the batch (printed by launch_read_verify): queued candidates, of which foreign /
// overhang, rounds of the lane path, candidates in them, passes, pieces, sketched reads, of which riding along
__device__ unsigned long long g_rv_stat[8];
__device__ __forceinline__ void rv_count(uint32_t debug, int which, uint32_t n, int lane)
{
    if ((debug & 1024u) && n && lane == 0) atomicAdd(&g_rv_stat[which], (unsigned long long)n);
}

// One pass of 64 pieces (61 evaluated) of the chunk's piece stream: bases -> canonical hashes + strands + minimizer bits, into the
// wave's LDS for the pieces of lanes 1..61.  Convergent: every lane runs it (the DPP shifts need all of them), lanes without a piece
// carry zeros.
template <int W, bool PACKED>
__device__ __forceinline__ void rv_sketch_pass(const SketchArgs& a, uint32_t pass, uint32_t n_pieces, uint32_t n_dense, int lane, RvWave& L)
{
    constexpr int K = 15;
    const int v = (int)(pass * RV_EVAL) - 1 + lane;
    const bool have = v >= 0 && v < (int)n_pieces;
    uint32_t le = 0, be = 0, diff = 0, validbits = 0, win = 0;
    uint4 in = make_uint4(0, 0, 0, 0);
    (void)in; (void)win;
    // the read of my piece: the last sketched read whose first piece is not behind it
    const uint32_t firsts = L.ld_pc0[lane];
    uint32_t didx = 0;
    for (uint32_t i = 1; i < n_dense; ++i) didx += rv_readlane(firsts, (int)i) <= (uint32_t)v ? 1u : 0u;
    if (have) {
        const uint32_t piece = (uint32_t)v - L.ld_pc0[didx];
        const int64_t r0 = (int64_t)L.ld_r0[didx], r1 = r0 + (int64_t)L.ld_len[didx];
        const int64_t g0 = (r0 & ~(int64_t)15) + 16 * (int64_t)piece; // global position of my first base: < r1 <= n_bases
        // k-mer starts g0 + j that are valid by position: r0 <= g0 + j <= r1 - K
        const int64_t lo = r0 - g0, hi = r1 - K - g0;
        if (hi >= 0) {
            const uint32_t upto = hi >= 15 ? 0xFFFFu : ((2u << (uint32_t)hi) - 1u);
            const uint32_t from = lo > 0 ? (0xFFFFu << (uint32_t)lo) & 0xFFFFu : 0xFFFFu;
            validbits = upto & from;
        }
        if constexpr (PACKED) {
            const uint32_t wd = reinterpret_cast<const uint32_t*>(a.bases)[g0 >> 4];
            sketch_pack_word(wd, le, be);
            // the positions of the batch that are not ACGT (ascending list; most batches have none): my 16 bases and the 16 behind them
            if (a.n_npos) win = (uint32_t)packed_bad_bases(a.npos, a.n_npos, g0);
            diff = win;
        } else {
            in = g0 + 16 <= (int64_t)a.n_bases ? *reinterpret_cast<const uint4*>(a.bases + g0) : load16_guarded(a.bases, (int64_t)a.n_bases, g0);
            sketch_pack_ascii(in, le, be, diff);
        }
    }
    le = sketch_letters_to_hash_order(le);
    be = sketch_letters_to_hash_order(be);
    const uint32_t le_next = from_next_lane(le), be_next = from_next_lane(be);
    if (__any(diff != 0)) { // rare: a k-mer that holds a base which is not ACGT is invalid
        if constexpr (!PACKED) {
            const uint32_t bad = have ? sketch_bad16(in) : 0u;
            win = bad | (from_next_lane(bad) << 16); // (a k-mer that runs into the next lane's bases is only valid if that lane holds the same read)
        }
        uint32_t inv = 0;
        for (int d = 0; d < K; ++d) inv |= win >> d; // k-mer j holds bases j .. j + K - 1
        validbits &= ~inv;
    }
    uint32_t hv[SB_G], strandbits;
    sketch_hashes16<K>(le, be, le_next, be_next, validbits, hv, strandbits);
    const uint32_t minbits = sketch_minimizers16<W>(hv) & validbits;
    if (have && lane >= 1 && lane <= RV_EVAL) {
        uint4* dst = L.hv4 + (size_t)(lane - 1) * 4;
        dst[0] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
        dst[1] = make_uint4(hv[4], hv[5], hv[6], hv[7]);
        dst[2] = make_uint4(hv[8], hv[9], hv[10], hv[11]);
        dst[3] = make_uint4(hv[12], hv[13], hv[14], hv[15]);
        L.bits[lane - 1] = minbits | (strandbits << 16);
    }
}

// what a slot keeps across the sketch: bit 31 = its read is sketched, [30:25] index among the sketched reads, [24:14] position in the
// read, [13:0] piece * 16 + position in the piece
__device__ __forceinline__ uint32_t rv_state(uint32_t didx, uint32_t pos, uint32_t vj) { return 0x80000000u | (didx << 25) | (pos << 14) | vj; }

template <int W, bool PACKED>
__global__ __launch_bounds__(RV_THREADS, DRPRG_RV_WAVES_PER_SIMD) void read_verify_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    constexpr int K = 15;
    __shared__ RvWave s_wave[RV_WAVES];
    __shared__ uint32_t s_red[3][RV_WAVES];
    __shared__ uint32_t s_m1[RV_WAVES * 64], s_m2[2 * RV_WAVES * 64], s_n1, s_n2; // the queues' leftovers, merged at the end
    if (threadIdx.x == 0) s_n1 = s_n2 = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    RvWave& L = s_wave[wv];
    const uint32_t total = *fw.cand_total;
    const VerifyConsts c(a, fw);
    uint32_t my_hits = 0, my_nmin = 0, my_maxlen = 0;
    uint32_t qn = 0, qn2 = 0; // entries waiting in the wave's two queues (wave-uniform)

    auto write_out = [&](uint32_t t, const VerifyOut& o) {
        fw.cand_pos1[t] = o.pos1;
        fw.cand_info[t] = ((uint64_t)o.slot << 32) | ((uint64_t)o.strand << 31) | (uint64_t)o.read;
        fw.cand_rec[t] = o.crec;
    };

    // The schedule: chunks round robin over the waves.  (Measured and not kept: cutting the end of the list -- the part that is less than one
    // chunk per wave -- into chunks of 64 candidates, so that no wave gets a whole chunk more than another: the small chunks cost more
    // instructions per candidate than they save in waiting, 122 us either way on the 8d index.)
    const uint32_t n_waves = gridDim.x * RV_WAVES, my_wave = blockIdx.x * RV_WAVES + (uint32_t)wv;
    const uint32_t n_chunks = (uint32_t)(((uint64_t)total + RV_OWN - 1) / RV_OWN);

    // one chunk: R owned rounds from list index `base`; prev_own: the owned size of the chunk before it
    auto process_chunk = [&](auto rounds_tag, const uint32_t base, const uint32_t prev_own) {
        constexpr int R = decltype(rounds_tag)::value;
        constexpr uint32_t OWN = 64u * R;
        {
            // (per-lane constants are made here, from a lane index the compiler cannot see through: hoisted out of the chunk loop they are
            // spilled to scratch memory and come back as VMEM loads)
            int ln0 = lane;
            uint32_t zero = 0;
            asm volatile("" : "+v"(ln0), "+v"(zero));
            const uint64_t lane_le = ln0 == 63 ? ~0ull : ((2ull << ln0) - 1ull); // bits 0 .. lane
            // ---- the chunk's positions (three owned rounds + the look-ahead), and the two neighbours that decide who owns what ----
            int64_t gp[R + 1];
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                const uint32_t t = base + 64u * (uint32_t)r + (uint32_t)lane;
                gp[r] = t < total ? (int64_t)fw.cand_gp[t] : -1;
            }
            const int64_t gp_prev = base ? (int64_t)fw.cand_gp[base - 1] : -1;                          // the candidate before this chunk
            const int64_t gp_prev2 = base > prev_own ? (int64_t)fw.cand_gp[base - prev_own - 1] : -1; // ... and before the previous chunk
            // ---- the read of every owned candidate (the interpolated index is exact for fixed-length reads; a short gallop otherwise) ----
            bool valid[R];
            uint32_t read[R];
            int64_t r0[R], r1[R];
            uint64_t o0[R], o1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { // (all six loads before the first is looked at)
                valid[r] = gp[r] >= c.win_lo && gp[r] < c.win_hi && gp[r] + K <= c.n_bases; // (gp = -1: past the end of the list)
                uint32_t guess = valid[r] ? (uint32_t)((double)gp[r] * c.reads_per_base) : 0u;
                if (guess >= a.n_reads) guess = a.n_reads - 1;
                read[r] = guess;
                o0[r] = a.offsets[guess];
                o1[r] = a.offsets[guess + 1];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                r0[r] = (int64_t)o0[r];
                r1[r] = (int64_t)o1[r];
                if (valid[r] && !(o0[r] <= (uint64_t)gp[r] && (uint64_t)gp[r] < o1[r])) {
                    read[r] = find_read_near(a.offsets, a.n_reads, read[r], (uint64_t)gp[r]);
                    r0[r] = (int64_t)a.offsets[read[r]];
                    r1[r] = (int64_t)a.offsets[read[r] + 1];
                }
                if (!valid[r]) read[r] = READ_NONE;
            }
            // ---- reads: a candidate whose predecessor lies in another read starts one ("leader") ----
            uint64_t Lm[R], Bm[R], Dm[R]; // leaders (Dm: of the reads that are sketched); leaders and slots without a candidate of this launch (where a read's run ends)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t before = from_prev_lane(read[r]);
                if (r > 0 && lane == 0) before = rv_readlane(read[r - 1], 63);
                const bool leader = valid[r] && (r == 0 && lane == 0 ? gp_prev < r0[r] : before != read[r]);
                Lm[r] = __ballot(leader);
                Bm[r] = __ballot(leader || !valid[r]);
            }
            // the look-ahead: the candidates behind the owned range that belong to the read of the last owned slot
            const bool last_valid = (__ballot(valid[R - 1]) >> 63) != 0;
            const int64_t r1_last = (int64_t)rv_readlane64((uint64_t)r1[R - 1], 63);
            const uint64_t Em = last_valid ? __ballot(gp[R] >= 0 && gp[R] < r1_last && gp[R] + K <= c.n_bases) : 0ull;
            const uint32_t n_ahead = (uint32_t)__popcll(Em);
            // ---- which reads are sketched, and where their pieces go ----
            uint32_t n_pieces = 0, n_dense = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool leader = ((Lm[r] >> lane) & 1ull) != 0;
                // the slot after the last candidate of my run: the next break, or the end of the owned range + the look-ahead's share
                const uint64_t after = lane == 63 ? 0ull : Bm[r] >> (lane + 1);
                uint32_t end = OWN + n_ahead;
                if (after) end = 64u * (uint32_t)r + (uint32_t)lane + 1u + (uint32_t)__builtin_ctzll(after);
                else {
#pragma unroll
                    for (int q = R - 1; q > r; --q)
                        if (Bm[q]) end = 64u * (uint32_t)q + (uint32_t)__builtin_ctzll(Bm[q]);
                }
                const uint32_t n_cand = end - (64u * (uint32_t)r + (uint32_t)lane);
                const int64_t len = r1[r] - r0[r];
                bool dense = leader && n_cand >= (uint32_t)RV_DENSE_MIN && len <= RV_MAX_LEN;
                const uint64_t dm0 = __ballot(dense);
                const uint32_t didx = n_dense + lanes_below(dm0);
                dense = dense && didx < (uint32_t)RV_MAXLEAD;
                const uint32_t n_pc = dense ? (uint32_t)(((r0[r] & 15) + len + 15) >> 4) : 0u;
                const uint32_t incl = wave_inclusive_scan(n_pc);
                const uint32_t pc0 = n_pieces + incl - n_pc;
                // (the state word holds piece * 16 + position in 14 bits: reads whose pieces would lie beyond that take the lane path -- from
                // the first such read of a round on, so that the stream of pieces has no holes)
                dense = dense && pc0 + n_pc <= (uint32_t)RV_MAXPIECES;
                const uint64_t dm = __ballot(dense);
                if (leader) L.s_lead[64 * r + lane] = dense ? (pc0 | (didx << 16)) : RV_NONE;
                if (dense) {
                    L.ld_r0[didx] = (uint64_t)r0[r];
                    L.ld_len[didx] = (uint32_t)len;
                    L.ld_read[didx] = read[r];
                    L.ld_pc0[didx] = pc0;
                }
                if (dm) n_pieces = rv_readlane(pc0 + n_pc, 63 - __builtin_clzll(dm));
                n_dense += (uint32_t)__popcll(dm);
                Dm[r] = dm;
            }
            // The last pass of a chunk is rarely full, and a lane costs nothing in a pass that runs anyway: reads with fewer candidates ride
            // along in the free lanes, in slot order, as long as they fit (measured before: their candidates went through the queue, and
            // every wave ended on a round of the lane path with half its lanes idle -- 4096 such rounds per batch, a third of all
            // instructions of the kernel)
            const uint32_t free_lanes = n_pieces ? (n_pieces + RV_EVAL - 1) / RV_EVAL * RV_EVAL - n_pieces : 0u;
            if (free_lanes >= 4u) {
                uint32_t used = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int64_t len = r1[r] - r0[r];
                    const bool cand = (((Lm[r] & ~Dm[r]) >> lane) & 1ull) != 0 && len <= RV_MAX_LEN;
                    const uint64_t cm = __ballot(cand);
                    if (!cm) continue;
                    const uint32_t n_pc = cand ? (uint32_t)(((r0[r] & 15) + len + 15) >> 4) : 0u;
                    const uint32_t incl = wave_inclusive_scan(n_pc);
                    const uint32_t didx = n_dense + lanes_below(cm);
                    const bool ok = cand && used + incl <= free_lanes && didx < (uint32_t)RV_MAXLEAD; // (a prefix of the round's candidates)
                    const uint64_t om = __ballot(ok);
                    if (ok) {
                        const uint32_t pc0 = n_pieces + used + incl - n_pc;
                        L.s_lead[64 * r + lane] = pc0 | (didx << 16);
                        L.ld_r0[didx] = (uint64_t)r0[r];
                        L.ld_len[didx] = (uint32_t)len;
                        L.ld_read[didx] = read[r];
                        L.ld_pc0[didx] = pc0;
                    }
                    if (om) used = rv_readlane(used + incl, 63 - __builtin_clzll(om));
                    n_dense += (uint32_t)__popcll(om);
                    rv_count(fw.debug, 7, (uint32_t)__popcll(om), lane);
                    if (om != cm) break; // (the first read that does not fit ends it: the pieces stay in slot order)
                }
                n_pieces += used;
            }
            if ((uint32_t)lane >= n_dense) L.ld_pc0[lane] = n_pieces;
            rv_fence();
            // ---- every candidate: the defaults, the queue, or -- after the sketch -- its own look-up (one word of state each: the sketch
            //      wants the registers) ----
            uint32_t state[R + 1];
            // the last leader before each round (1 + slot; 0: none)
            uint32_t lead_before = 0;
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                const uint32_t t = base + 64u * (uint32_t)r + (uint32_t)lane;
                state[r] = 0;
                bool queue = false, foreign = false;
                if (r < R) {
                    const uint64_t upto = Lm[r] & lane_le;
                    const uint32_t lead = upto ? 64u * (uint32_t)r + 64u - (uint32_t)__builtin_clzll(upto) : lead_before; // 1 + slot of my read's leader
                    if (gp[r] >= 0) {
                        if (!valid[r]) { // (outside this launch's read range: the defaults)
                            fw.cand_pos1[t] = zero;
                            fw.cand_info[t] = (uint64_t)(READ_NONE + zero);
                            fw.cand_rec[t] = make_uint4(zero, zero, zero, zero);
                        }
                        else if (!lead) {
                            // my read began before this chunk: the chunk before handled me iff the read began in ITS owned range and I lie in its look-ahead
                            const bool theirs = r == 0 && (base <= prev_own || gp_prev2 < r0[r]);
                            queue = !theirs;
                            foreign = queue;
                        } else {
                            const uint32_t ld = L.s_lead[lead - 1];
                            if (ld == RV_NONE) queue = true;
                            else {
                                const uint32_t pos = (uint32_t)(gp[r] - r0[r]);
                                state[r] = rv_state(ld >> 16, pos, (ld & 0xFFFFu) * 16u + (uint32_t)(r0[r] & 15) + pos);
                            }
                        }
                    }
                    if (Lm[r]) lead_before = 64u * (uint32_t)r + 64u - (uint32_t)__builtin_clzll(Lm[r]);
                } else if (((Em >> lane) & 1ull) && lead_before) { // (no leader: the read began before this chunk, its look-ahead is the next chunk's)
                    const uint32_t ld = L.s_lead[lead_before - 1];
                    if (ld == RV_NONE) queue = true;
                    else {
                        const uint32_t didx = ld >> 16;
                        const int64_t lr0 = (int64_t)L.ld_r0[didx];
                        const uint32_t pos = (uint32_t)(gp[r] - lr0);
                        state[r] = rv_state(didx, pos, (ld & 0xFFFFu) * 16u + (uint32_t)(lr0 & 15) + pos);
                    }
                }
                const uint64_t qm = __ballot(queue);
                if (queue) L.q[qn + lanes_below(qm)] = t;
                qn += (uint32_t)__popcll(qm);
                rv_count(fw.debug, 0, (uint32_t)__popcll(qm), lane);
                if (fw.debug & 1024u) rv_count(fw.debug, 1, (uint32_t)__popcll(__ballot(foreign)), lane);
            }
            rv_count(fw.debug, 4, (n_pieces + RV_EVAL - 1) / RV_EVAL, lane);
            rv_count(fw.debug, 5, n_pieces, lane);
            rv_count(fw.debug, 6, n_dense, lane);
            // ---- the sketch, pass by pass; behind every pass the candidates whose piece it evaluated look themselves up ----
            for (uint32_t p = 0; p * RV_EVAL < n_pieces && !(fw.debug & 128u); ++p) {
                rv_sketch_pass<W, PACKED>(a, p, n_pieces, n_dense, lane, L);
                rv_fence();
#pragma unroll
                for (int r = 0; r <= R; ++r) {
                    // (the state word and the lane index enter every look-up as values the compiler knows nothing about: otherwise everything
                    // that derives from them -- fields, LDS addresses, three output addresses per round -- is computed before the pass loop
                    // and kept across the sketch, which wants every register: 200 bytes of scratch per lane)
                    uint32_t st = state[r];
                    int ln = lane;
                    asm volatile("" : "+v"(st), "+v"(ln));
                    const uint32_t vj = st & 0x3FFFu;
                    const uint32_t pv = (vj >> 4) - p * (uint32_t)RV_EVAL; // piece within the pass
                    const bool mine = (st >> 31) && pv < (uint32_t)RV_EVAL && !(fw.debug & 256u);
                    if (!__any(mine)) continue;
                    if (mine) {
                        const uint32_t t = base + 64u * (uint32_t)r + (uint32_t)ln;
                        const uint32_t didx = (st >> 25) & 63u, pos = (st >> 14) & 0x7FFu, j = vj & 15u;
                        VerifyOut o;
                        o.read = L.ld_read[didx];
                        const uint32_t bits = L.bits[pv];
                        if ((bits >> j) & 1u) { // a minimizer of its read: is it an index k-mer?
                            const uint32_t h = reinterpret_cast<const uint32_t*>(L.hv4)[pv * 16 + j] - 1u;
                            o.strand = (bits >> (16 + j)) & 1u;
                            uint32_t sl = table_slot_dev(h, a.table_bits);
                            bool found = false;
                            while (true) {
                                const uint32_t key = c.slot_key[sl];
                                if (key == h) { found = true; break; }
                                if (key == HashTraits<uint32_t>::EMPTY) break;
                                sl = (sl + 1) & c.tmask;
                            }
                            o.slot = sl;
                            if (found) {
                                const uint4 sf = a.slot_first[sl];
                                verify_emit(a, rc, c, (int64_t)pos, 0, (int64_t)L.ld_len[didx], o.strand, sf, o, my_hits, my_nmin, my_maxlen); // (position and length are all it takes)
                            }
                        }
                        write_out(t, o);
                    }
                }
                rv_fence(); // (the next pass overwrites the hashes)
            }
        }
    };

    for (uint32_t chunk = my_wave; chunk < n_chunks; chunk += n_waves) {
        process_chunk(std::integral_constant<int, RV_ROUNDS>(), chunk * (uint32_t)RV_OWN, (uint32_t)RV_OWN);
        // ---- the lane path, in two steps with a queue each: first the lookup alone (most candidates of stray reads are false positives
        //      of the Bloom filter and end there), then, 64 index k-mers at a time, the window scan ----
        while (qn >= 64u) {
            qn -= 64u;
            rv_count(fw.debug, 2, 1u, lane);
            if (!(fw.debug & 64u)) { // (DRPRG_FT_DEBUG=64 / 128 / 256: timing only -- no lane path / sketch / look-up)
                const uint32_t t = L.q[qn + (uint32_t)lane];
                const int64_t gpq = (int64_t)fw.cand_gp[t];
                VerifyOut o;
                const bool pass = probe_one_lane<K, PACKED>(a, c, gpq, o);
                if (!pass) write_out(t, o);
                const uint64_t pm = __ballot(pass);
                if (pass) L.q2[qn2 + lanes_below(pm)] = t;
                qn2 += (uint32_t)__popcll(pm);
            }
            rv_fence();
            if (qn2 >= 64u) {
                qn2 -= 64u;
                rv_count(fw.debug, 3, 1u, lane);
                const uint32_t t = L.q2[qn2 + (uint32_t)lane];
                const int64_t gpq = (int64_t)fw.cand_gp[t];
                VerifyOut o;
                verify_one_lane<K, PACKED>(a, fw, rc, c, gpq, o, my_hits, my_nmin, my_maxlen);
                write_out(t, o);
                rv_fence();
            }
        }
    }
    // ---- what is left in the queues of the workgroup's four waves goes through the two steps together: a round of the window scan
    //      costs the same with 8 lanes busy as with 64 ----
    {
        uint32_t b1 = 0, b2 = 0;
        if (lane == 0) {
            b1 = atomicAdd(&s_n1, qn);
            b2 = atomicAdd(&s_n2, qn2);
        }
        b1 = rv_readlane(b1, 0);
        b2 = rv_readlane(b2, 0);
        if ((uint32_t)lane < qn) s_m1[b1 + (uint32_t)lane] = L.q[lane];
        if ((uint32_t)lane < qn2) s_m2[b2 + (uint32_t)lane] = L.q2[lane];
    }
    __syncthreads();
    {
        const uint32_t n1 = s_n1, i = 64u * (uint32_t)wv + (uint32_t)lane; // (n1 <= 4 * 63)
        __syncthreads(); // (everybody has read s_n1 ... s_n2 grows from here)
        bool pass = false;
        uint32_t t = 0;
        if (i < n1 && !(fw.debug & 64u)) {
            t = s_m1[i];
            VerifyOut o;
            pass = probe_one_lane<K, PACKED>(a, c, (int64_t)fw.cand_gp[t], o);
            if (!pass) write_out(t, o);
        }
        const uint64_t pm = __ballot(pass);
        uint32_t b2 = 0;
        if (pm && lane == 0) b2 = atomicAdd(&s_n2, (uint32_t)__popcll(pm));
        b2 = rv_readlane(b2, 0);
        if (pass) s_m2[b2 + lanes_below(pm)] = t;
    }
    __syncthreads();
    {
        const uint32_t n2 = s_n2; // (<= 8 * 63)
        for (uint32_t i = 64u * (uint32_t)wv + (uint32_t)lane; i - (uint32_t)lane < n2; i += 64u * RV_WAVES) {
            if (i < n2) {
                const uint32_t t = s_m2[i];
                VerifyOut o;
                verify_one_lane<K, PACKED>(a, fw, rc, c, (int64_t)fw.cand_gp[t], o, my_hits, my_nmin, my_maxlen);
                write_out(t, o);
            }
        }
    }
    // ---- per-workgroup totals ----
    const uint32_t wh = wave_inclusive_scan(my_hits), wn = wave_inclusive_scan(my_nmin), wm = wave_max(my_maxlen);
    if (lane == 63) {
        s_red[0][wv] = wh;
        s_red[1][wv] = wn;
        s_red[2][wv] = wm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t h = 0, n = 0, mx = 0;
        for (int i = 0; i < RV_WAVES; ++i) {
            h += s_red[0][i];
            n += s_red[1][i];
            mx = s_red[2][i] > mx ? s_red[2][i] : mx;
        }
        fw.wg_hits[blockIdx.x] = h;
        fw.wg_nmin[blockIdx.x] = n;
        fw.wg_maxlen[blockIdx.x] = mx;
    }
}

// DRPRG_VERIFY_FORM=read (read per call) selects this kernel for the batches it serves; the default stays verify_count_kernel, which it
// does not beat (header of this file)
bool read_verify_applies(const SketchArgs& a, const FilterWork& fw)
{
    const char* form = std::getenv("DRPRG_VERIFY_FORM");
    if (!(form && std::string(form) == "read")) return false;
    if (a.k != 15 || (a.w != 11 && a.w != 14) || a.n_reads == 0) return false;
    if (a.n_bases / a.n_reads > 300) return false; // long reads: hundreds of candidates per read, the look-ahead of a chunk does not hold them
    if (fw.debug & (16u | 32u)) return false;      // (verify_count_kernel's ablation switches; 64 ... 1024 are this kernel's)
    return true;
}

uint32_t read_verify_grid(int n_cus) { return (uint32_t)n_cus * (uint32_t)DRPRG_RV_WAVES_PER_SIMD; } // persistent: the workgroups of four waves that stay resident at 128 VGPRs

hipError_t launch_read_verify(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc, uint32_t grid, hipStream_t stream)
{
    const dim3 g(grid), b(RV_THREADS);
    if (a.packed) {
        if (a.w == 11) hipLaunchKernelGGL((read_verify_kernel<11, true>), g, b, 0, stream, a, fw, rc);
        else hipLaunchKernelGGL((read_verify_kernel<14, true>), g, b, 0, stream, a, fw, rc);
    } else {
        if (a.w == 11) hipLaunchKernelGGL((read_verify_kernel<11, false>), g, b, 0, stream, a, fw, rc);
        else hipLaunchKernelGGL((read_verify_kernel<14, false>), g, b, 0, stream, a, fw, rc);
    }
    if (fw.debug & 1024u) { // (debugging only: waits for the kernel)
        unsigned long long h[8] = {}, z[8] = {};
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rv_stat), sizeof h));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rv_stat), z, sizeof z));
        std::fprintf(stderr, "[read_verify] queued %llu (foreign/overhang %llu) lookup rounds %llu, scan rounds %llu (without the merged leftovers); passes %llu pieces %llu sketched reads %llu (riding along %llu)\n",
            h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
