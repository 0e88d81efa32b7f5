// verify_lane.h -- one candidate k-mer verified by one lane, start to finish: the device functions verify_count_kernel
// (candidates.hip) is made of, shared with read_verify_kernel (read_verify.hip), which sends the candidates of sparsely hit reads
// through them.  Exact index lookup (pandora add_read_hits) + "is it a (w,k) window minimizer of its read" (Seq::minimizer_sketch)
// inside the external `pandora map` process of /root/reference/src/lib.rs:580-642.
#pragma once
#include "filter_common.h"
#ifndef DRPRG_VERIFY_OUTWARD
#define DRPRG_VERIFY_OUTWARD 1
#endif

namespace drprg {
namespace dev {

// 16 ASCII bases -> packed codes (A0 C1 G2 T3, first base highest) + 16-bit "not ACGT" mask (bit i = base i)
__device__ inline void pack16n(const uint4& in, uint32_t& packed, uint32_t& nmask)
{
    // four bases: their codes (encode4's v_perm_b32 look-up), and what tells a byte that is no base -- it differs from the letter its index
    // stands for.  The per-byte flags encode4 makes of that difference (six more instructions per dword) are left to the rare case below
    auto diff_of = [](uint32_t word, uint32_t& sel) {
        const uint32_t u = word & 0xDFDFDFDFu; // upper case
        sel = (u >> 1) & 0x03030303u;          // A0 C1 T2 G3
        return u ^ __builtin_amdgcn_perm(0u, 0x47544341u, sel);
    };
    uint32_t any = 0;
    auto enc = [&](uint32_t word) {
        uint32_t sel;
        any |= diff_of(word, sel);
        return __builtin_amdgcn_perm(0u, 0x02030100u, sel); // -> A0 C1 G2 T3
    };
    const uint32_t c0 = enc(in.x), c1 = enc(in.y), c2 = enc(in.z), c3 = enc(in.w);
    // v_dot4_u32_u8 with the byte weights 64, 16, 4, 1: the four codes of a dword in one byte, first base highest (full rate;
    // the 32 x 32 multiply that gathers them is quarter rate)
    constexpr uint32_t W = 0x01041040u;
    packed = (__builtin_amdgcn_udot4(c0, W, 0u, false) << 24) | (__builtin_amdgcn_udot4(c1, W, 0u, false) << 16) | (__builtin_amdgcn_udot4(c2, W, 0u, false) << 8)
        | __builtin_amdgcn_udot4(c3, W, 0u, false);
    nmask = 0;
    if (any) { // rare; bit 7 of every non-zero byte, then (flags * 0x01020408) >> 24 gathers the flag of byte i into bit i
        auto m4 = [&](uint32_t word) {
            uint32_t sel;
            const uint32_t diff = diff_of(word, sel);
            const uint32_t bad = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
            return (((bad >> 7) * 0x01020408u) >> 24) & 0xFu;
        };
        nmask = m4(in.x) | (m4(in.y) << 4) | (m4(in.z) << 8) | (m4(in.w) << 12);
    }
}

// reverse complement of a k-mer code (2 bits per base, k <= 16)
__device__ inline uint32_t revcomp_code(uint32_t f, int k)
{
    uint32_t x = __brev(f);                                    // bit reversal also swaps the two bits of every base
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);   // swap them back
    return (~x) >> (32 - 2 * k);                               // complement, right-align
}

// 16 bases at global position g (a multiple of 16); bytes past the end of the buffer read as 'N'
__device__ inline uint4 load16_guarded(const uint8_t* __restrict__ bases, int64_t n_bases, int64_t g)
{
    if (g + 16 <= n_bases) return *reinterpret_cast<const uint4*>(bases + g);
    uint32_t t4[4];
    for (int q = 0; q < 4; ++q) {
        uint32_t wd = 0;
        for (int b = 0; b < 4; ++b) {
            const int64_t gg = g + q * 4 + b;
            wd |= (uint32_t)(gg < n_bases ? bases[gg] : (uint8_t)'N') << (8 * b);
        }
        t4[q] = wd;
    }
    return make_uint4(t4[0], t4[1], t4[2], t4[3]);
}

// 16 bases of a packed batch (SketchArgs::packed: letters A0 C1 T2 G3, first base in the lowest bits) -> the form pack16n makes of 16
// ASCII bases: A0 C1 G2 T3, first base highest.  letter ^ (letter >> 1) swaps T and G; v_bfrev reverses the order of the sixteen
// 2-bit fields and the two bits of each, which are then swapped back.
__device__ __forceinline__ uint32_t packed_to_hash_order(uint32_t x)
{
    const uint32_t y = x ^ ((x >> 1) & 0x55555555u);
    const uint32_t r = __brev(y);
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}

// bit i: base a0 + i of a packed batch is not one of ACGTacgt (npos: ascending positions; most batches have none)
__device__ inline uint64_t packed_bad_bases(const uint64_t* __restrict__ npos, uint64_t n_npos, int64_t a0)
{
    uint64_t lo = 0, hi = n_npos;
    while (lo < hi) { // first position >= a0
        const uint64_t mid = (lo + hi) >> 1;
        if ((int64_t)npos[mid] < a0) lo = mid + 1; else hi = mid;
    }
    uint64_t bad = 0;
    for (; lo < n_npos && (int64_t)npos[lo] < a0 + 64; ++lo) bad |= 1ull << ((int64_t)npos[lo] - a0);
    return bad;
}

// One lane per candidate, start to finish: the 64 bases around it are packed into registers once (2 bits per base,
// first base highest), the candidate's canonical hash goes to the exact table lookup (Bloom false positives end there),
// then its neighbours are hashed out of three of those words by the same lane -- a third of the instructions of giving every
// neighbour its own lane, each of which had to load and pack its own bases.  It is a read minimizer iff the run of neighbours
// with hash >= its own (inside the read, no N) reaches w-1 across both sides: the lane walks outward from the candidate, left
// until the first smaller hash, then right as far as still needed -- at most w hashes (rounds 1-4: all 2w-1 in turn).
// KC: compile-time k (15: the mask-free 12-instruction hash mix_k) or 0 (any k <= 15)
template <int KC> __device__ __forceinline__ uint32_t verify_mix(uint32_t x, uint32_t kmask)
{
    if constexpr (KC > 0) return mix_k<KC>(x);
    else return HashTraits<uint32_t>::mix(x, kmask);
}

// what the kernels below derive once from their arguments
struct VerifyConsts {
    const uint32_t* __restrict__ slot_key;
    uint32_t tmask, kmask, w1_magic;
    int k, w, sh_k;
    int64_t n_bases, win_lo, win_hi;
    double reads_per_base;
    __device__ VerifyConsts(const SketchArgs& a, const FilterWork& fw)
        : slot_key(reinterpret_cast<const uint32_t*>(a.slot_key)), tmask((1u << a.table_bits) - 1), kmask((1u << (2 * a.k)) - 1), w1_magic(w1_reciprocal(a.w)),
          k(a.k), w(a.w), sh_k(32 - 2 * a.k), n_bases((int64_t)a.n_bases), win_lo((int64_t)a.offsets[fw.read_begin]), win_hi((int64_t)a.offsets[fw.read_end]),
          reads_per_base((double)a.n_reads / (double)(a.n_bases ? a.n_bases : 1))
    {
    }
};
// everything one candidate leaves behind
struct VerifyOut {
    uint32_t pos1 = 0, slot = 0, read = READ_NONE, strand = 0;
    uint4 crec = make_uint4(0, 0, 0, 0);
};

// the record of a candidate that is a minimizer of its read: what read_cluster_kernel needs of it (and the lane's totals)
__device__ __forceinline__ void verify_emit(const SketchArgs& a, const ReadClusterArgs& rc, const VerifyConsts& c, int64_t gp, int64_t r0, int64_t r1, uint32_t strand,
    const uint4& sf, VerifyOut& o, uint32_t& my_hits, uint32_t& my_nmin, uint32_t& my_maxlen)
{
    const uint64_t pos = (uint64_t)(gp - r0);
    if (pos >= (1ull << HIT_POS_BITS)) {
        atomicOr(a.overflow, 2u);
        return;
    }
    o.pos1 = (uint32_t)pos + 1;
    my_hits += sf.y;
    my_nmin += 1;
    const uint32_t len = (uint32_t)((r1 - r0) > 0xFFFFFFFFll ? 0xFFFFFFFFll : (r1 - r0));
    my_maxlen = len > my_maxlen ? len : my_maxlen;
    // for read_cluster_kernel: the first hit of this minimizer and the size threshold of a cluster of this read on that hit's PRG
    // (cluster_eval_kernel)
    const uint32_t kn = sf.z, prg = sf.w & 0xFFFu;
    const uint32_t rev = ((kn & 1u) == strand) ? 0u : 1u;
    const uint64_t expected = expected_minimizers((uint64_t)(r1 - r0), c.w, c.w1_magic);
    uint64_t m = sf.w >> 12;
    if (expected < m) m = expected;
    const uint32_t length_based = (uint32_t)((double)m * rc.fraction);
    uint32_t thr = length_based > rc.min_cluster_size ? length_based : rc.min_cluster_size;
    if (thr > 0xFFFFu) thr = 0xFFFFu; // read_cluster_kernel stages at most RC_HCAP hits: no difference
    o.crec = make_uint4(sf.x, sf.y, (strand << 31) | (((prg << 1) | rev) << 16) | thr, (kn >> 1) * 2u + rev);
}

// The first half of verify_one_lane on its own: the read of a candidate, its canonical hash, the exact lookup -- and nothing of the window
// scan.  true: the k-mer is an index k-mer that lies inside its read (verify_one_lane then decides whether it is a minimizer); false: o
// is what verify_one_lane would leave (no minimizer; the read).  read_verify_kernel sends the candidates of sparsely hit reads through
// this first: most of them are false positives of the Bloom filter, which end here at a sixth of the cost.
// gp: a candidate position of this launch's read range (win_lo <= gp < win_hi, gp + k <= n_bases)
template <int KC, bool PACKED>
__device__ __forceinline__ bool probe_one_lane(const SketchArgs& a, const VerifyConsts& c, int64_t gp, VerifyOut& o)
{
    const int k = c.k;
    const int64_t n_bases = c.n_bases;
    uint32_t guess = (uint32_t)((double)gp * c.reads_per_base);
    if (guess >= a.n_reads) guess = a.n_reads - 1;
    const uint64_t o0 = a.offsets[guess], o1 = a.offsets[guess + 1];
    const int64_t a0 = gp & ~(int64_t)15; // the k-mer lies in the 32 bases from here
    uint32_t r0w, r1w, bad;
    if constexpr (PACKED) {
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(a.bases) + (a0 >> 4);
        const uint32_t x0 = wp[0], x1 = a0 + 16 < n_bases ? wp[1] : 0u;
        r0w = packed_to_hash_order(x0);
        r1w = packed_to_hash_order(x1);
        bad = a.n_npos ? (uint32_t)packed_bad_bases(a.npos, a.n_npos, a0) : 0u;
    } else {
        const uint4 b0 = a0 + 16 <= n_bases ? *reinterpret_cast<const uint4*>(a.bases + a0) : load16_guarded(a.bases, n_bases, a0);
        const uint4 b1 = a0 + 32 <= n_bases ? *reinterpret_cast<const uint4*>(a.bases + a0 + 16) : load16_guarded(a.bases, n_bases, a0 + 16);
        uint32_t n0, n1;
        pack16n(b0, r0w, n0);
        pack16n(b1, r1w, n1);
        bad = n0 | (n1 << 16);
    }
    int64_t r1 = (int64_t)o1;
    if (o0 <= (uint64_t)gp && (uint64_t)gp < o1) o.read = guess;
    else {
        o.read = find_read_near(a.offsets, a.n_reads, guess, (uint64_t)gp);
        r1 = (int64_t)a.offsets[o.read + 1];
    }
    const int oc = (int)(gp - a0); // 0..15: the k-mer is bases oc .. oc + k - 1 <= 29 of the 32
    if ((bad >> oc) & ((1u << k) - 1u)) return false; // a base that is not ACGT
    const uint32_t f = __funnelshift_l(r1w, r0w, 2 * oc) >> c.sh_k;
    const uint32_t hf = verify_mix<KC>(f, c.kmask), hr = verify_mix<KC>(revcomp_code(f, k), c.kmask);
    o.strand = hf <= hr ? 1u : 0u;
    const uint32_t h = hf < hr ? hf : hr;
    uint32_t sl = table_slot_dev(h, a.table_bits);
    const bool found = table_find4(c.slot_key, c.tmask, h, sl);
    o.slot = sl;
    return found && gp + k <= r1;
}

// One candidate, start to finish, by one lane (the whole algorithm described above verify_count_kernel)
template <int KC, bool PACKED>
__device__ __forceinline__ void verify_one_lane(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc, const VerifyConsts& c, int64_t gp, VerifyOut& o,
    uint32_t& my_hits, uint32_t& my_nmin, uint32_t& my_maxlen)
{
    const uint32_t* __restrict__ slot_key = c.slot_key;
    const uint32_t tmask = c.tmask, kmask = c.kmask;
    const int k = c.k, w = c.w, sh_k = c.sh_k;
    const int64_t n_bases = c.n_bases, win_lo = c.win_lo, win_hi = c.win_hi;
    const double reads_per_base = c.reads_per_base;
    uint32_t &pos1 = o.pos1, &slot = o.slot, &read = o.read, &strand = o.strand;
    (void)pos1;
    if (gp >= win_lo && gp < win_hi && gp + k <= n_bases) { // (the boundary tiles of a read range reach past it)
        // Everything the candidate needs from memory that does not depend on other loads is requested before anything is
        // waited for: the two read offsets around the interpolated read index and the four 16-byte words of the 64 bases
        // [a0, a0+64) that hold the candidate and all its neighbours (w <= 16, k <= 15).  (One guarded load after the other,
        // each behind its own branch, was four round trips in a row, and the read lookup two more.)
        uint32_t guess = (uint32_t)((double)gp * reads_per_base);
        if (guess >= a.n_reads) guess = a.n_reads - 1;
        const uint64_t o0 = a.offsets[guess], o1 = a.offsets[guess + 1];
        const int64_t a0 = (gp > 15 ? gp - 15 : 0) & ~(int64_t)15;
        uint4 b0, b1, b2, b3;
        uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0; // packed input: the four words of [a0, a0 + 64)
        (void)b0; (void)b1; (void)b2; (void)b3; (void)x0; (void)x1; (void)x2; (void)x3;
        if constexpr (PACKED) {
            const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(a.bases) + (a0 >> 4);
            const int64_t left = ((n_bases + 15) >> 4) - (a0 >> 4); // words from a0 on: >= 1 (gp + k <= n_bases)
            x0 = wp[0];
            if (left >= 4) {
                x1 = wp[1];
                x2 = wp[2];
                x3 = wp[3];
            } else {
                if (left > 1) x1 = wp[1];
                if (left > 2) x2 = wp[2];
            }
        } else if (a0 + 64 <= n_bases) {
            const uint4* __restrict__ bp = reinterpret_cast<const uint4*>(a.bases + a0);
            b0 = bp[0];
            b1 = bp[1];
            b2 = bp[2];
            b3 = bp[3];
        } else { // the last bytes of the buffer
            b0 = load16_guarded(a.bases, n_bases, a0);
            b1 = load16_guarded(a.bases, n_bases, a0 + 16);
            b2 = load16_guarded(a.bases, n_bases, a0 + 32);
            b3 = load16_guarded(a.bases, n_bases, a0 + 48);
        }
        // the read of every candidate, index k-mer or not: read_cluster_kernel finds the first candidate of a read by
        // comparing neighbours (the interpolated index is exact for fixed-length reads; a short gallop otherwise)
        int64_t r0 = (int64_t)o0, r1 = (int64_t)o1;
        if (o0 <= (uint64_t)gp && (uint64_t)gp < o1) read = guess;
        else {
            read = find_read_near(a.offsets, a.n_reads, guess, (uint64_t)gp);
            r0 = (int64_t)a.offsets[read];
            r1 = (int64_t)a.offsets[read + 1];
        }
        uint32_t r0w, r1w, r2w, r3w;
        uint64_t bad; // bit i: base a0+i is not ACGT (or lies behind the last base of the batch)
        if constexpr (PACKED) {
            r0w = packed_to_hash_order(x0);
            r1w = packed_to_hash_order(x1);
            r2w = packed_to_hash_order(x2);
            r3w = packed_to_hash_order(x3);
            bad = a.n_npos ? packed_bad_bases(a.npos, a.n_npos, a0) : 0ull;
            if (a0 + 64 > n_bases) bad |= ~0ull << (n_bases - a0);
        } else {
            uint32_t n0, n1, n2, n3;
            pack16n(b0, r0w, n0);
            pack16n(b1, r1w, n1);
            pack16n(b2, r2w, n2);
            pack16n(b3, r3w, n3);
            bad = (uint64_t)(n0 | (n1 << 16)) | ((uint64_t)(n2 | (n3 << 16)) << 32);
        }
        if (bad) { // -> bit i: the k-mer starting at a0+i holds such a base
            uint64_t m = bad;
            for (int i = 1; i < k; ++i) m |= bad >> i;
            bad = m;
        }
        // ---- the candidate's own canonical hash, exact lookup ----
        const int oc = (int)(gp - a0); // 0..30
        uint32_t g = 0;
        if (!((bad >> oc) & 1u)) {
            const uint32_t h0 = (oc & 16) ? r1w : r0w, h1 = (oc & 16) ? r2w : r1w;
            const uint32_t f = __funnelshift_l(h1, h0, 2 * (oc & 15)) >> sh_k;
            const uint32_t hf = verify_mix<KC>(f, kmask), hr = verify_mix<KC>(revcomp_code(f, k), kmask);
            strand = hf <= hr ? 1u : 0u;
            g = (hf < hr ? hf : hr) + 1;
        }
        bool found = false;
        if (g && !(fw.debug & 32u)) { // (DRPRG_FT_DEBUG=32: timing only, no table probe and nothing after it)
            const uint32_t h = g - 1;
            uint32_t sl = table_slot_dev(h, a.table_bits);
            found = table_find4(slot_key, tmask, h, sl);
            slot = sl;
        }
        if (found) {
            if (gp + k <= r1) { // the k-mer lies inside one read
                // (requested before the window scan that decides whether it is needed: the scan hides the round trip)
                const uint4 sf = a.slot_first[slot]; // record offset, count, first record's node, its prg and that prg's shortest path
                // ---- scan q = q_first .. q_first + 2w-2; steps outside [gp-(w-1), gp+(w-1)] or the read are invalid ----
                const int64_t q_lo = gp - (w - 1);
                const int64_t q_first = q_lo > a0 ? q_lo : a0; // a0 <= max(q_lo, 0)
                const int of = (int)(q_first - a0);            // 0..30
                const int64_t v_lo = r0 > q_first ? r0 : q_first;
                const int64_t v_hi = (r1 - k) < (gp + w - 1) ? (r1 - k) : (gp + w - 1);
                const int i_lo = (int)(v_lo - q_first), i_hi = (int)(v_hi - q_first), ic = (int)(gp - q_first);
                // align the shift register on q_first: 48 bases in three words cover 2w-1 + k-1 <= 45
                if (of & 16) { r0w = r1w; r1w = r2w; r2w = r3w; r3w = 0; }
                const int s2 = 2 * (of & 15);
                r0w = __funnelshift_l(r1w, r0w, s2);
                r1w = __funnelshift_l(r2w, r1w, s2);
                r2w = __funnelshift_l(r3w, r2w, s2);
                // steps that can count at all: inside the read and the window, no N in the k-mer (bit i = step i)
                const uint32_t valid = ((2u << i_hi) - 1u) & ~((1u << i_lo) - 1u) & ~(uint32_t)(bad >> of);
#if DRPRG_VERIFY_OUTWARD
                // The candidate is a minimizer iff the neighbours with hash >= its own form a run of w-1 around it: L on its left, then
                // w-1-L on its right.  Walked outward from the candidate -- left until the first smaller hash (or enough), then right
                // exactly as far as still needed -- that is at most w hashes per lane, whatever L is, where every step q_first ..
                // q_first + 2w-2 in turn is 2w-1.  (Each lane extracts the k-mer of ITS step from the three words; the reverse
                // complement is formed per step instead of rolled.)  What bounds the runs before any hash: the steps that can count.
                const uint32_t below = ic ? (valid << (32 - ic)) : 0u;         // bit 31 = step ic-1
                const int l_cap = (int)__builtin_clz(~below | (ic ? 0u : 0x80000000u)); // ones from bit 31 down (ic = 0: none)
                const uint32_t above = valid >> (ic + 1);                     // bit 0 = step ic+1
                const int r_cap = (int)__builtin_ctz(~above);                 // (bits 30.. of `above` are clear)
                const int need = w - 1;
                const int l_lim = l_cap < need ? l_cap : need;
                int dir = l_lim > 0 ? -1 : 1, j = ic + dir, remaining = need; // remaining: neighbours still to be shown >= the candidate
                bool going = l_cap + r_cap >= need && need > 0 && (dir < 0 || need <= r_cap) && !(fw.debug & 16u);
                bool is_min = need == 0 || (fw.debug & 16u) != 0; // (DRPRG_FT_DEBUG=16: measurement only, no window test)
                // (the three words one bit to the right: step j then starts at the odd bit 2j + 1, and the funnel shift that extracts it is
                // never one by zero -- which v_alignbit_b32 cannot do and the compiler guards with a branch)
                const uint32_t q0 = r0w >> 1, q1 = __builtin_amdgcn_alignbit(r0w, r1w, 1), q2 = __builtin_amdgcn_alignbit(r1w, r2w, 1);
                while (__builtin_amdgcn_ballot_w64(going)) { // (lanes that are through compute along: nothing of theirs is kept)
                    const uint32_t bo = 2u * (uint32_t)j + 1u;
                    const uint32_t hi = (bo & 32u) ? q1 : q0, lo = (bo & 32u) ? q2 : q1;
                    const uint32_t f = __builtin_amdgcn_alignbit(hi, lo, 0u - bo) >> sh_k; // ((hi:lo) << (bo & 31)) >> 32: the shift count is taken mod 32
                    const uint32_t hf = verify_mix<KC>(f, kmask), hr = verify_mix<KC>(revcomp_code(f, k), kmask);
                    const bool ok = (hf < hr ? hf : hr) + 1 >= g;
                    remaining -= ok ? 1 : 0;
                    const bool was_left = dir < 0;
                    const bool turn = was_left & (!ok | (ic - j == l_lim)); // the left run ends here: the right one has to bring the rest
                    const bool dead = (turn & (remaining > r_cap)) | (!was_left & !ok); // (plain & and |: lane masks, no selects)
                    is_min = is_min | (going & (remaining == 0));
                    going = going & (remaining != 0) & !dead;
                    dir = turn ? 1 : dir;
                    j = turn ? ic + 1 : j + dir;
                }
                const uint32_t streak = is_min ? (uint32_t)need : 0u, right = 0;
#else
                uint32_t streak = 0, right = 0, alive = 1;
                uint32_t rcw = revcomp_code(r0w >> sh_k, k) << 2; // the reverse complement rolls along: one base in, one out
                const int n_steps = (fw.debug & 16u) ? 0 : 2 * w - 1; // (DRPRG_FT_DEBUG=16: measurement only, no window test)
                for (int i = 0; i < n_steps; ++i) {
                    const uint32_t f = r0w >> sh_k;
                    rcw = (rcw >> 2) | ((~f & 3u) << (2 * k - 2));
                    r0w = __funnelshift_l(r1w, r0w, 2);
                    r1w = __funnelshift_l(r2w, r1w, 2);
                    r2w <<= 2;
                    const uint32_t hf = verify_mix<KC>(f, kmask), hr = verify_mix<KC>(rcw, kmask);
                    const uint32_t x = (hf < hr ? hf : hr) + 1;
                    const bool ok = ((valid >> i) & 1u) && x >= g;
                    if (i < ic) streak = ok ? streak + 1 : 0;
                    else if (i > ic) {
                        alive = ok ? alive : 0u;
                        right += alive;
                    }
                }
#endif
                if ((int)(streak + right) >= w - 1) verify_emit(a, rc, c, gp, r0, r1, strand, sf, o, my_hits, my_nmin, my_maxlen);
            }
        }
    }
}

} // namespace dev
} // namespace drprg
