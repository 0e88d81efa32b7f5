// read_verify.hip -- the verification stage of the filtered launch sequence for short-read batches, READ BY READ (round 5).
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
// index).  This kernel answers it per READ: a workgroup stages 256 consecutive candidates, finds their reads (the candidates of a
// read sit next to each other), and every read that holds at least RV_DENSE_MIN of them is sketched ONCE -- its bases laid out 16
// per lane over as many lanes as it needs, reads back to back, hashed and window-minimised in registers by the block
// sketch_wave_kernel is made of (sketch_block.h; neighbours by DPP wave shifts) -- after which each candidate only looks its own
// position up (minimizer bit, canonical hash, strand: LDS), probes the exact table if it is a minimizer and writes its record.
// Reads with fewer candidates (an off-panel read that happens to share one 15-mer with the index: 1 % of all reads, as many reads
// as come from the panel) would waste a sketch: their candidates are queued and go through verify_one_lane (verify_lane.h) 256 at
// a time, all lanes busy.  Same outputs as verify_count_kernel, bit for bit where it matters downstream (cand_pos1, the read of
// every candidate, slot / strand / record of every minimizer, the per-workgroup totals).
//
// Ownership.  Chunk c = candidates [c OWN, c OWN + SLOTS) of the list, OWN = SLOTS - LOOK: the last LOOK slots are look-ahead.  A
// read belongs to the chunk whose OWNED range holds its first candidate; that chunk handles the read's candidates up to the end of
// its staged range, the ones beyond ("overhang": a read with more than LOOK candidates) are handled one by one, through the queue,
// by the chunk that owns their list index.  Every decision is taken from positions alone (fw.cand_gp is never written here), so
// every candidate is written by exactly one workgroup.
//
// The sketch of a chunk.  The dense reads of a chunk are laid out as one stream of 16-base pieces ("virtual chunks": piece i of a
// read = its bases [A + 16 i, A + 16 i + 16), A = the read's start rounded down to 16, so that every piece is one aligned 16-byte
// load -- or one word of a 2-bit packed batch).  Wave p evaluates pieces [61 p, 61 p + 61) in its lanes 1..61; lane 0 and lanes
// 62, 63 hold the pieces before and after and only supply neighbours (the same tiling as sketch_wave_kernel), so a read may run
// from one pass into the next.  Reads never see each other's k-mers: a k-mer is valid iff it starts at or after its read's first base
// and ends inside the read, an invalid k-mer is 0, a window holding one has minimum 0, and the last K-1 >= 1 positions of every
// read are invalid.
#include "filter_common.h"
#include "sketch_block.h"
#include "verify_lane.h"
#include <cstdlib>
#include <string>

namespace drprg {
namespace dev {

constexpr int RV_THREADS = 256;
constexpr int RV_WAVES = RV_THREADS / 64;
constexpr int RV_SLOTS = RV_THREADS;          // staged candidates: one per thread
constexpr int RV_LOOK = 64;                   // of which look-ahead
constexpr int RV_OWN = RV_SLOTS - RV_LOOK;
constexpr int RV_EVAL = 61;                   // pieces a pass evaluates (lanes 1..61)
constexpr int RV_VCAP = RV_WAVES * RV_EVAL;   // pieces per chunk at most: one pass per wave (reads beyond it take the lane path)
constexpr int RV_DENSE_MIN = 5;               // candidates a read must hold in the chunk to be sketched: a sketch costs a read of 150
                                              // bases ~11 lanes x 900 / 61 wave-instructions, the lane path ~30 per candidate
constexpr int RV_MAX_LEN = 1024;              // longer reads (stray ones in a short-read batch) take the lane path
constexpr int RV_QCAP = 2 * RV_THREADS;       // queue of candidates for the lane path (drained whenever a full round is waiting)
constexpr uint32_t RV_NONE = 0xFFFFu;

// One pass of 64 pieces (61 evaluated): bases -> canonical hashes + strands + minimizer bits, into LDS for the pieces of lanes 1..61.
// Convergent: every lane of the wave runs it (the DPP shifts need all of them), lanes without a piece carry zeros.
template <int W, bool PACKED>
__device__ __forceinline__ void rv_sketch_pass(const SketchArgs& a, uint32_t pass, uint32_t n_pieces, int lane, const uint32_t* s_vmap, const uint64_t* s_r0,
    const uint32_t* s_len, uint4* s_hv4, uint32_t* s_bits)
{
    constexpr int K = 15;
    const int v = (int)(pass * RV_EVAL) - 1 + lane;
    const bool have = v >= 0 && v < (int)n_pieces;
    uint32_t le = 0, be = 0, diff = 0, validbits = 0, win = 0;
    uint4 in = make_uint4(0, 0, 0, 0);
    (void)in; (void)win;
    if (have) {
        const uint32_t e = s_vmap[v];
        const uint32_t leader = e & 0xFFFFu, piece = e >> 16;
        const int64_t r0 = (int64_t)s_r0[leader], r1 = r0 + (int64_t)s_len[leader];
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
        uint4* dst = s_hv4 + (size_t)v * 4;
        dst[0] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
        dst[1] = make_uint4(hv[4], hv[5], hv[6], hv[7]);
        dst[2] = make_uint4(hv[8], hv[9], hv[10], hv[11]);
        dst[3] = make_uint4(hv[12], hv[13], hv[14], hv[15]);
        s_bits[v] = minbits | (strandbits << 16);
    }
}

template <int W, bool PACKED>
__global__ __launch_bounds__(RV_THREADS, 5) void read_verify_kernel(SketchArgs a, FilterWork fw, ReadClusterArgs rc)
{
    constexpr int K = 15;
    __shared__ uint4 s_hv4[RV_VCAP * 4];   // per evaluated piece: canonical hash + 1 of its 16 positions (0 = invalid)
    __shared__ uint32_t s_bits[RV_VCAP];   // per evaluated piece: minimizer bits | strand bits << 16
    __shared__ uint32_t s_vmap[RV_VCAP];   // piece -> slot of its read's first candidate | piece number in the read << 16
    __shared__ uint64_t s_r0[RV_SLOTS];    // at a read's first slot: first base of the read ...
    __shared__ uint32_t s_len[RV_SLOTS];   // ... and its length
    __shared__ uint32_t s_read[RV_SLOTS];  // per slot: read (READ_NONE: not a candidate of this launch's read range)
    __shared__ uint16_t s_end[RV_SLOTS];   // at a read's first slot: the slot after its last staged candidate
    __shared__ uint16_t s_vbase[RV_SLOTS]; // at a read's first slot: its first piece (RV_NONE: the read takes the lane path)
    __shared__ uint32_t s_qt[RV_QCAP];     // the queue for the lane path: list indices
    __shared__ uint32_t s_w[RV_WAVES + 1], s_lastlead[RV_WAVES];
    __shared__ uint32_t s_red[3][RV_WAVES];
    __shared__ uint32_t s_pieces, s_qn;

    int tid = threadIdx.x;
#define lane (tid & 63)
#define wave (tid >> 6)
    const uint32_t total = *fw.cand_total;
    const VerifyConsts c(a, fw);
    uint32_t my_hits = 0, my_nmin = 0, my_maxlen = 0;
    uint32_t qn = 0; // entries waiting in the queue (the same value in every thread; s_qn while a chunk adds to it)

    auto write_out = [&](uint32_t t, const VerifyOut& o) {
        fw.cand_pos1[t] = o.pos1;
        fw.cand_info[t] = ((uint64_t)o.slot << 32) | ((uint64_t)o.strand << 31) | (uint64_t)o.read;
        fw.cand_rec[t] = o.crec;
    };
    auto drain = [&](uint32_t from, uint32_t n) { // queue entries [from, from + n), n <= RV_THREADS: one lane each
        if ((uint32_t)tid < n) {
            const uint32_t t = s_qt[from + (uint32_t)tid];
            const int64_t gp = (int64_t)fw.cand_gp[t];
            VerifyOut o;
            verify_one_lane<K, PACKED>(a, fw, rc, c, gp, o, my_hits, my_nmin, my_maxlen);
            write_out(t, o);
        }
    };

    for (uint64_t base64 = (uint64_t)blockIdx.x * RV_OWN; base64 < total; base64 += (uint64_t)gridDim.x * RV_OWN) {
        // (the thread index comes into every chunk as a value the compiler knows nothing about: what derives from it -- LDS addresses, lane
        // masks, shuffle indices -- is computed where it is used instead of once before the loop, where it filled the registers, was
        // spilled to scratch memory and came back as VMEM loads inside the loop: 170 VGPRs wanted, 96 there at five waves per SIMD)
        asm volatile("" : "+v"(tid));
        const uint32_t base = (uint32_t)base64;
        const uint32_t n_loaded = total - base < (uint32_t)RV_SLOTS ? total - base : (uint32_t)RV_SLOTS;
        const uint32_t n_own = total - base < (uint32_t)RV_OWN ? total - base : (uint32_t)RV_OWN;
        const uint32_t t = base + (uint32_t)tid;
        const bool in = (uint32_t)tid < n_loaded;
        // ---- the chunk's positions, and the two neighbours that decide who owns what ----
        const int64_t gp = in ? (int64_t)fw.cand_gp[t] : -1;
        const int64_t gp_prev = base ? (int64_t)fw.cand_gp[base - 1] : -1;                          // the candidate before this chunk
        const int64_t gp_prev2 = base > (uint32_t)RV_OWN ? (int64_t)fw.cand_gp[base - RV_OWN - 1] : -1; // ... and before the previous chunk
        // ---- the read of every candidate (the interpolated index is exact for fixed-length reads; a short gallop otherwise) ----
        const bool valid = in && gp >= c.win_lo && gp < c.win_hi && gp + K <= c.n_bases;
        uint32_t read = READ_NONE;
        int64_t r0 = 0, r1 = 0;
        if (valid) {
            uint32_t guess = (uint32_t)((double)gp * c.reads_per_base);
            if (guess >= a.n_reads) guess = a.n_reads - 1;
            const uint64_t o0 = a.offsets[guess], o1 = a.offsets[guess + 1];
            if (o0 <= (uint64_t)gp && (uint64_t)gp < o1) {
                read = guess;
                r0 = (int64_t)o0;
                r1 = (int64_t)o1;
            } else {
                read = find_read_near(a.offsets, a.n_reads, guess, (uint64_t)gp);
                r0 = (int64_t)a.offsets[read];
                r1 = (int64_t)a.offsets[read + 1];
            }
        }
        s_read[tid] = read;
        if (tid == 0) s_pieces = 0;
        __syncthreads();
        if (tid == 0) s_qn = qn; // (everybody has read the previous chunk's count; nobody adds before the third barrier from here)
        // ---- reads: a candidate whose predecessor lies in another read starts one ("leader"); lead = 1 + the slot of my read's leader,
        //      0 if my read began before this chunk ----
        const bool leader = valid && (tid == 0 ? gp_prev < r0 : s_read[tid - 1] != read);
        const uint64_t lm = __ballot(leader);
        const uint64_t lm_upto = lm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        uint32_t lead = lm_upto ? (uint32_t)(wave * 64 + 63 - __clzll((long long)lm_upto)) + 1u : 0u;
        if (lane == 0) s_lastlead[wave] = lm ? (uint32_t)(wave * 64 + 63 - __clzll((long long)lm)) + 1u : 0u;
        __syncthreads();
        if (!lead)
            for (int i = wave - 1; i >= 0; --i)
                if (s_lastlead[i]) {
                    lead = s_lastlead[i];
                    break;
                }
        if (valid && lead && ((uint32_t)tid + 1 >= n_loaded || s_read[tid + 1] != read)) s_end[lead - 1] = (uint16_t)(tid + 1);
        __syncthreads();
        // ---- which reads are sketched, and where their pieces go ----
        uint32_t n_pc = 0;
        const bool my_read = leader && (uint32_t)tid < n_own;
        if (my_read) {
            const uint32_t n_cand = (uint32_t)s_end[tid] - (uint32_t)tid;
            s_r0[tid] = (uint64_t)r0;
            s_len[tid] = (uint32_t)(r1 - r0 > 0xFFFFFFFFll ? 0xFFFFFFFFll : r1 - r0);
            if (n_cand >= (uint32_t)RV_DENSE_MIN && r1 - r0 <= RV_MAX_LEN) n_pc = (uint32_t)((r1 - (r0 & ~(int64_t)15) + 15) >> 4);
        }
        uint32_t sum;
        const uint32_t first_pc = block_exclusive_scan<RV_WAVES>(n_pc, s_w, &sum);
        if (my_read) {
            const bool dense = n_pc != 0 && first_pc + n_pc <= (uint32_t)RV_VCAP;
            s_vbase[tid] = dense ? (uint16_t)first_pc : (uint16_t)RV_NONE;
            if (dense) {
                for (uint32_t i = 0; i < n_pc; ++i) s_vmap[first_pc + i] = (uint32_t)tid | (i << 16);
                atomicMax(&s_pieces, first_pc + n_pc);
            }
        }
        __syncthreads();
        // ---- every candidate: the defaults, the queue, or -- after the sketch -- its own look-up.  Everything the look-up needs is four
        //      words (the sketch wants the registers) ----
        bool dense = false, queue = false;
        uint32_t vj = 0, pos = 0, len = 0;
        if (in) {
            if (!valid) {
                if ((uint32_t)tid < n_own) write_out(t, VerifyOut()); // (outside this launch's read range)
            } else if (!lead) {
                // my read began before this chunk: the chunk before handled me iff the read began in ITS owned range and I lie in its look-ahead
                const bool theirs = tid < RV_LOOK && (base <= (uint32_t)RV_OWN || gp_prev2 < r0);
                queue = (uint32_t)tid < n_own && !theirs;
            } else if (lead <= n_own) {
                const uint32_t first_pc_of_read = s_vbase[lead - 1];
                if (first_pc_of_read == RV_NONE) queue = true;
                else {
                    dense = true;
                    pos = (uint32_t)(gp - r0);
                    len = (uint32_t)(r1 - r0);
                    vj = first_pc_of_read * 16u + pos + (uint32_t)(r0 & 15);
                }
            } // (else: the read begins in the look-ahead: the next chunk's)
        }
        {
            const uint64_t qm = __ballot(queue);
            uint32_t q0 = 0;
            if (qm && lane == 0) q0 = atomicAdd(&s_qn, (uint32_t)__popcll(qm));
            q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q0);
            if (queue) s_qt[q0 + lanes_below(qm)] = t;
        }
        // ---- the sketch: one pass of 61 pieces per wave ----
        const uint32_t n_pieces = s_pieces;
        for (uint32_t p = (uint32_t)wave; p * RV_EVAL < n_pieces; p += RV_WAVES)
            rv_sketch_pass<W, PACKED>(a, p, n_pieces, lane, s_vmap, s_r0, s_len, s_hv4, s_bits);
        __syncthreads();
        qn = s_qn;
        if (dense) {
            VerifyOut o;
            o.read = read;
            const uint32_t v = vj >> 4, j = vj & 15u;
            const uint32_t bits = s_bits[v];
            if ((bits >> j) & 1u) { // a minimizer of its read: is it an index k-mer?
                const uint32_t h = reinterpret_cast<const uint32_t*>(s_hv4)[vj] - 1u;
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
                    verify_emit(a, rc, c, (int64_t)pos, 0, (int64_t)len, o.strand, sf, o, my_hits, my_nmin, my_maxlen); // (position and length are all it takes)
                }
            }
            write_out(t, o);
        }
        if (qn >= (uint32_t)RV_THREADS) {
            qn -= (uint32_t)RV_THREADS;
            drain(qn, (uint32_t)RV_THREADS);
        }
    }
    __syncthreads();
    if (qn) drain(0, qn);
    // ---- per-workgroup totals ----
    const uint32_t wh = wave_inclusive_scan(my_hits), wn = wave_inclusive_scan(my_nmin), wm = wave_max(my_maxlen);
    if (lane == 63) {
        s_red[0][wave] = wh;
        s_red[1][wave] = wn;
        s_red[2][wave] = wm;
    }
    __syncthreads();
    if (tid == 0) {
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
#undef lane
#undef wave
}

// DRPRG_VERIFY_FORM=lane keeps verify_count_kernel for every batch (A/B runs, and a second way through the parity tests); read per call
bool read_verify_applies(const SketchArgs& a, const FilterWork& fw)
{
    if (a.k != 15 || (a.w != 11 && a.w != 14) || a.n_reads == 0) return false;
    if (a.n_bases / a.n_reads > 300) return false; // long reads: hundreds of candidates per read, the look-ahead of a chunk does not hold them
    if (fw.debug & (16u | 32u)) return false;      // (verify_count_kernel's ablation switches)
    const char* form = std::getenv("DRPRG_VERIFY_FORM");
    return !(form && std::string(form) == "lane");
}

uint32_t read_verify_grid(int n_cus) { return (uint32_t)n_cus * 5u; } // persistent: what stays resident at ~96 VGPRs

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
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
