// sketch_wave.hip -- K1+K2 for large indexes, register-resident form (gfx950): sketch_wave_kernel<K, W>.
//
// Same contract as the candidate form of sketch_probe_kernel (sketch_probe.hip): every k-mer of the concatenated base buffer
// is hashed, the (w,k) window minimizers are found, the ones that are index keys leave the workgroup as candidate records in
// position order in its slice (tile_info / tile_pos1 / tile_rec, tile_count / tile_hits / tile_nmin); scan + gather +
// read_cluster_kernel take it from there.  It replaces, inside the external `pandora map` process that
// /root/reference/src/lib.rs:580-642 spawns, Seq::minimizer_sketch + the index lookup of add_read_hits.
//
// What is different from sketch_probe_kernel: nothing goes through LDS on the way from bases to window minima, and no
// workgroup barrier exists -- every WAVE owns a tile of 64 x 16 bases:
//   * one 16-byte non-temporal load per lane; the 16 bases are packed to 2 bits twice (first base in the low bits / in the
//     high bits) with v_dot4_u32_u8; the right neighbour's words arrive by a DPP wave shift;
//   * k-mer j of a lane is ONE v_alignbit_b32 out of the high-first words (forward k-mer) and ONE out of the low-first words
//     (whose complement is the reverse-complement k-mer, so that costs nothing): no rolling, no masks;
//   * the hash is mix_k<K> (12 full-rate instructions, device_common.h); canonical hash + 1 (0 = invalid k-mer) stays in 16
//     VGPRs; the W-1 values either side come from the neighbouring lanes by DPP wave shifts;
//   * window minima by doubling in registers, as in sketch_probe_kernel;
//   * lanes 0, 62 and 63 only supply neighbours (a tile evaluates 61 x 16 positions);
//   * read boundaries and non-ACGT bases become a per-lane mask of invalid k-mer starts (a 64-word LDS array per wave, touched
//     only by that wave);
//   * the minimizers of the tile are compacted per wave (prefix over lanes with wave scans) and thinned in passes with all
//     lanes busy: Bloom tier -> exact table lookup -> reads that lie inside the tile are clustered on the spot and their
//     coverage added (stage C1) -> a record for what is left; the four tiles of a workgroup share one slice.
#include "sketch_block.h"

namespace drprg {
namespace dev {

constexpr int SW_G = SB_G;                  // k-mer positions (= bytes loaded) per lane (sketch_block.h)
constexpr int SW_LOAD = 64 * SW_G;          // bases staged per wave tile
constexpr int SW_FIRST = 1, SW_LAST = 61;   // lanes that evaluate their positions
constexpr int SW_EVAL = (SW_LAST - SW_FIRST + 1) * SW_G; // 976 positions per tile
constexpr int SW_WAVES = 4;                 // independent waves (= tiles) per workgroup

uint32_t wave_tile_eval() { return SW_EVAL; }
uint32_t wave_n_tiles(uint64_t n_bases) { return (uint32_t)((n_bases + SW_EVAL - 1) / SW_EVAL); }
bool wave_kernel_applies(int k, int w) { return k == 15 && (w == 11 || w == 14); }
uint32_t wave_n_slices(uint64_t n_bases) { return (wave_n_tiles(n_bases) + SW_WAVES - 1) / SW_WAVES; } // one slice per workgroup

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Volatile accesses to a wave's LDS go through pointers that SAY they point into LDS: a volatile access through a generic pointer
// is left alone by the address-space inference and comes out as flat_load ... sc0 sc1 + s_waitcnt vmcnt(0) -- the slow path into
// LDS, and a wait that also drains every global load in flight (the three Bloom words of stage A were requested "in one round
// trip" and fetched one after the other).
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
typedef volatile __attribute__((address_space(3))) uint16_t lds_vu16;
#define LDS_U32(p) ((lds_vu32*)(p))
#define LDS_U16(p) ((lds_vu16*)(p))

// LDS of one wave (nothing in it is shared between waves)
struct alignas(16) WaveLds {
    uint32_t hv[SW_LOAD];   // hash + 1 of every position of the wave's tile (0 = invalid); later the slot of a found one
    uint16_t list[SW_LOAD]; // tile positions: minimizers -> those that pass the Bloom tier -> index keys (compacted in place)
    uint32_t inv[64];       // per lane: k-mer starts invalidated by a read boundary
    uint32_t strand[64];    // per lane: bit j = forward k-mer of position j is the canonical one
    uint32_t rbits[64];     // per lane: bit j = a read starts at position j; bit 16 = several reads start at one position
    uint32_t rcnt[64];      // per lane: reads that start in its 16 positions; later the read before its first position
};

template <int K, int W, bool FUSE, bool PACKED>
__device__ __forceinline__ void sketch_wave_tile(const SketchArgs& a, uint32_t tile, bool active, WaveLds& lds, uint32_t* s_nb, uint32_t* s_sum, int lane, int wave)
{
    if (!active) { // a wave past the last tile still meets the workgroup's two barriers
        (void)s_sum;
        if (lane == 0) s_nb[wave] = 0;
        __syncthreads();
        __syncthreads();
        return;
    }
    const int64_t n_bases = (int64_t)a.n_bases;
    const int64_t origin = (int64_t)tile * SW_EVAL - SW_G; // global position of lane 0's first base
    const int64_t g0 = origin + (int64_t)lane * SW_G;

    // ---- bases -> 2-bit words, low-first (le) and high-first (be) ----
    lds.inv[lane] = 0;
    lds.rbits[lane] = 0;
    lds.rcnt[lane] = 0;
    uint32_t le, be, diff = 0;
    uint4 in = make_uint4(0, 0, 0, 0);
    uint32_t bad16 = 0; // PACKED: bit i = my base i is not ACGT (listed in npos) or lies outside the batch
    (void)in; (void)bad16;
    if constexpr (PACKED) {
        // a lane's 16 bases are one word of the batch (g0 is a multiple of 16); the letters are the low-first stream as they are, the
        // high-first stream is the word with its sixteen fields in reverse order (v_bfrev + the two bits of every field swapped back)
        const uint32_t* const words = reinterpret_cast<const uint32_t*>(a.bases);
        uint32_t wd = 0;
        if (g0 >= 0 && g0 < n_bases) {
            wd = words[g0 >> 4];
            if (a.nbits) bad16 = a.nbits[g0 >> 4];
            if (g0 + SW_G > n_bases) bad16 |= (0xFFFFu << (uint32_t)(n_bases - g0)) & 0xFFFFu;
        } else {
            bad16 = 0xFFFFu;
        }
        sketch_pack_word(wd, le, be);
        diff = bad16;
    } else {
        if (g0 >= 0 && g0 + SW_G <= n_bases) {
            in = load_once_16(a.bases + g0);
        } else { // the ends of the buffer: what lies outside is 'N'
            uint32_t tmp[4];
            for (int q = 0; q < 4; ++q) {
                uint32_t wd = 0;
                for (int b = 0; b < 4; ++b) {
                    const int64_t gg = g0 + q * 4 + b;
                    wd |= (uint32_t)((gg >= 0 && gg < n_bases) ? a.bases[gg] : (uint8_t)'N') << (8 * b);
                }
                tmp[q] = wd;
            }
            in = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
        sketch_pack_ascii(in, le, be, diff);
    }
    le = sketch_letters_to_hash_order(le); // A0 C1 T2 G3 -> A0 C1 G2 T3 in every 2-bit field
    be = sketch_letters_to_hash_order(be);
    const uint32_t le_next = from_next_lane(le), be_next = from_next_lane(be);

    // ---- read boundaries: a k-mer must not straddle two reads (invalid k-mer starts), and where reads start (read lookup) ----
    const uint32_t first_read = a.tile_first_read[tile];
    {
        const int64_t end_pos = origin + SW_LOAD;
        for (uint32_t r = first_read + (uint32_t)lane; r < a.n_reads; r += 64) {
            const int64_t o = (int64_t)a.offsets[r];
            if (o >= end_pos + K) break;
            const int oc = (int)(o - origin); // the k-mers that start in [oc - K + 1, oc) end inside read r
            if (oc >= 0 && oc < SW_LOAD) {
                atomicOr(&lds.rbits[oc >> 4], 1u << (oc & 15));
                atomicAdd(&lds.rcnt[oc >> 4], 1u);
            }
            int p0 = oc - K + 1, p1 = oc < SW_LOAD ? oc : SW_LOAD;
            if (p0 < 0) p0 = 0;
            if (p0 >= p1) continue;
            const int l0 = p0 >> 4, l1 = (p1 - 1) >> 4; // at most two lanes (K - 1 <= 16 positions)
            const uint32_t run = ((1u << (p1 - p0)) - 1u) << (p0 & 15);
            atomicOr(&lds.inv[l0], run & 0xFFFFu);
            if (l1 != l0) atomicOr(&lds.inv[l1], run >> 16);
        }
    }
    // ---- ... and bases that are not ACGT (rare: the exact per-base test only runs when some lane saw one) ----
    uint32_t inv = 0;
    if (__any(diff != 0)) {
        uint32_t bad;
        if constexpr (PACKED) {
            bad = bad16;
        } else {
            bad = sketch_bad16(in);
        }
        const uint32_t win = bad | (from_next_lane(bad) << 16); // my 16 bases and the 16 after them
        for (int d = 0; d < K; ++d) inv |= win >> d;           // k-mer j holds bases j .. j + K - 1
    }
    wave_lds_fence();
    inv |= *LDS_U32(&lds.inv[lane]);
    const uint32_t validbits = ~inv & 0xFFFFu;
    { // read that holds the base just before my first position = first_read - 1 + reads that start before it in this tile
        const uint32_t c = *LDS_U32(&lds.rcnt[lane]);
        const uint32_t rb = *LDS_U32(&lds.rbits[lane]);
        const uint32_t before = wave_inclusive_scan(c) - c;
        if (c != (uint32_t)__popc(rb)) lds.rbits[lane] = rb | 0x10000u; // empty reads: the popcount shortcut does not hold here
        lds.rcnt[lane] = first_read - 1u + before;
    }

    // ---- canonical hash + 1 of my 16 k-mers (0 = invalid), strand bits ----
    uint32_t hv[SW_G];
    uint32_t strandbits;
    sketch_hashes16<K>(le, be, le_next, be_next, validbits, hv, strandbits);

    // ---- window minimizers (sketch_block.h); lanes 0, 62 and 63 only supply neighbours ----
    uint32_t minbits = sketch_minimizers16<W>(hv) & validbits;
    if (lane < SW_FIRST || lane > SW_LAST) minbits = 0;

    // ---- the tile's minimizers, compacted in position order; hashes and strands parked in LDS for the probe loops ----
    {
        uint4* dst = reinterpret_cast<uint4*>(&lds.hv[lane * SW_G]);
        dst[0] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
        dst[1] = make_uint4(hv[4], hv[5], hv[6], hv[7]);
        dst[2] = make_uint4(hv[8], hv[9], hv[10], hv[11]);
        dst[3] = make_uint4(hv[12], hv[13], hv[14], hv[15]);
        lds.strand[lane] = strandbits;
    }
    const uint32_t cnt = (uint32_t)__popc(minbits);
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t nmin = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    {
        uint32_t at = incl - cnt, bits = minbits;
        while (bits) {
            const int j = __ffs(bits) - 1;
            bits &= bits - 1;
            lds.list[at++] = (uint16_t)(lane * SW_G + j);
        }
    }
    wave_lds_fence();
    lds_vu16* list = LDS_U16(lds.list);
    lds_vu32* hvs = LDS_U32(lds.hv);

    // ---- stage A: the Bloom tier in front of a table that outgrows the L2 stops most minimizers at one word ----
    uint32_t n_a = nmin;
    if (a.pbloom) {
        n_a = 0;
        // (three rounds of 64 minimizers at a time -- a tile has about 160 --, their three filter words requested before the first
        // one is looked at: one round trip to the L2 instead of three)
        for (uint32_t i0 = 0; i0 < nmin; i0 += 192) {
            uint32_t p[3] = { 0, 0, 0 }, need[3] = { 0, 0, 0 }, word[3] = { 0, 0, 0 };
            bool in[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
                in[u] = i < nmin;
                p[u] = list[in[u] ? i : 0u];
                const uint32_t m = (hvs[p[u]] - 1u) * 0x9E3779B1u; // (= pbloom_mix of a 32-bit key)
                need[u] = pbloom_bits(m);
                word[u] = a.pbloom[pbloom_word(m, a.pbloom_wbits)];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool pass = in[u] && (word[u] & need[u]) == need[u];
                const uint64_t pm = __ballot(pass);
                if (pass) list[n_a + lanes_below(pm)] = (uint16_t)p[u]; // (in place: nothing at or beyond the entries read above is overwritten)
                n_a += (uint32_t)__popcll(pm);
            }
        }
        wave_lds_fence();
    }
    // ---- stage B: exact lookup; the slot of a found key replaces its hash, the list shrinks to the index keys ----
    const uint32_t tmask = (1u << a.table_bits) - 1;
    const uint32_t* __restrict__ slot_key = reinterpret_cast<const uint32_t*>(a.slot_key);
    uint32_t n_b = 0;
    for (uint32_t i0 = 0; i0 < n_a; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        bool found = false;
        uint32_t p = 0, s = 0;
        if (i < n_a) {
            p = list[i];
            const uint32_t h = hvs[p] - 1u;
            s = table_slot_dev(h, a.table_bits);
            found = table_find4(slot_key, tmask, h, s); // (four slots per round trip)
        }
        const uint64_t fm = __ballot(found);
        if (found) {
            list[n_b + lanes_below(fm)] = (uint16_t)p;
            hvs[p] = s;
        }
        n_b += (uint32_t)__popcll(fm);
    }
    wave_lds_fence();
    // everything a record, or a hit of the in-kernel clustering, needs of list entry i
    struct Entry {
        uint32_t p, slot, read, strand, kn, prg, rev, thr;
        uint4 sf; // record offset, count, the first record's node << 1 | strand, its prg | shortest path << 12
        uint64_t o0, o1, pos;
    };
    const uint32_t w1_magic = w1_reciprocal(W);
    auto decode = [&](uint32_t i) -> Entry {
        Entry e;
        e.p = list[i] & 0x7FFFu;
        e.slot = hvs[e.p];
        const uint64_t gp = (uint64_t)(origin + (int64_t)e.p);
        const uint32_t rb = *LDS_U32(&lds.rbits[e.p >> 4]);
        e.read = *LDS_U32(&lds.rcnt[e.p >> 4]) + (uint32_t)__popc(rb & ((2u << (e.p & 15)) - 1u));
        bool ok = !(rb & 0x10000u) && e.read < a.n_reads;
        // (the slot's record and the read's two offsets in one round trip: the offsets unconditionally, of read 0 if the count is off)
        const uint32_t rd = ok ? e.read : 0u;
        e.sf = a.slot_first[e.slot];
        e.o0 = a.offsets[rd];
        e.o1 = a.offsets[rd + 1];
        ok = ok && e.o0 <= gp && gp < e.o1;
        if (!ok) { // empty reads around here: search from the tile's first read
            e.read = find_read_from(a.offsets, a.n_reads, first_read ? first_read - 1 : 0, gp);
            e.o0 = a.offsets[e.read];
            e.o1 = a.offsets[e.read + 1];
        }
        e.pos = gp - e.o0;
        e.strand = (*LDS_U32(&lds.strand[e.p >> 4]) >> (e.p & 15)) & 1u;
        e.kn = e.sf.z;
        e.prg = e.sf.w & 0xFFFu;
        e.rev = ((e.kn & 1u) == e.strand) ? 0u : 1u;
        // size threshold of a cluster of this read on that PRG: floor(min(shortest path, expected) * fraction) =
        // min(floor(shortest path * fraction) [per PRG, from the host], floor(expected * fraction)) -- floor(x * f) is monotone
        const uint32_t by_len = (uint32_t)((double)expected_minimizers(e.o1 - e.o0, W, w1_magic) * a.fraction);
        const uint32_t by_prg = a.prg_thr[e.prg];
        const uint32_t length_based = by_len < by_prg ? by_len : by_prg;
        e.thr = length_based > a.min_cluster_size ? length_based : a.min_cluster_size;
        if (e.thr > 0xFFFFu) e.thr = 0xFFFFu;
        return e;
    };

    // ---- stage C1: reads that lie inside this tile are clustered here (pandora define_clusters / filter_clusters /
    // add_hits_to_kmergraphs, as read_cluster_kernel's segment path does them).  Such a read has every one of its index
    // minimizers in this wave's list, next to each other and in position order.  If each has exactly one index record and
    // all of them fall into ONE (prg, strand) group, its clusters are the runs without a position gap > max_diff, a cluster is
    // kept iff it has more hits than the size threshold, and the overlap sweep cannot drop a kept one (same group, disjoint
    // ranges): every hit of a kept cluster adds 1 to its k-mer node's coverage right here.  Every other read -- it crosses the
    // tile's edge, a minimizer with several records, hits in two groups, more than 64 index minimizers -- keeps its entries
    // for stage C2 (records -> gather -> read_cluster_kernel).  a.fuse: 0 off (the default: see Mapper), 1 on, -1 take the same
    // additions back (the host's undo pass before it re-runs a batch whose record slices overflowed).
    uint32_t n_f = n_b, fast_clusters = 0, fast_hits = 0, my_hits = 0;
    if constexpr (FUSE) {
        n_f = 0;
        uint32_t cont_read = 0xFFFFFFFFu; // a read with more entries than one pass holds: never clustered here
        for (uint32_t c = 0; c < n_b;) {
            const uint32_t i = c + (uint32_t)lane;
            const bool valid = i < n_b;
            Entry e {};
            if (valid) e = decode(i);
            const uint32_t r_prev = from_prev_lane(e.read), pos_prev = from_prev_lane((uint32_t)e.pos);
            const bool head = valid && (lane == 0 || e.read != r_prev);
            const uint64_t hm = __ballot(head);
            uint32_t take = n_b - c < 64u ? n_b - c : 64u;
            bool oversized = false;
            if (c + 64 < n_b) { // entries follow: the last read of this pass may run on -- leave it to the next pass
                const int last_head = 63 - __clzll((long long)hm);
                if (last_head > 0) take = (uint32_t)last_head;
                else oversized = true; // one read fills the pass
            }
            const bool active = valid && (uint32_t)lane < take;
            const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull); // bits 0 .. lane
            // my read: entries [rs, re)
            const int rs = 63 - __clzll((long long)((hm & upto) | 1ull));
            const uint64_t h_after = lane == 63 ? 0ull : (hm >> (lane + 1));
            uint32_t re = h_after ? (uint32_t)lane + (uint32_t)__ffsll((long long)h_after) : 64u;
            if (re > take) re = take;
            const uint32_t g = (e.prg << 1) | e.rev;
            const uint32_t g0 = (uint32_t)__shfl((int)g, rs);
            const int64_t s_loc = (int64_t)e.o0 - origin, e_loc = (int64_t)e.o1 - origin;
            const bool owned = s_loc >= SW_FIRST * SW_G && e_loc <= (SW_LAST + 1) * SW_G - 1 + K; // every k-mer of the read starts in an evaluated lane
            // a minimizer with several index records (the same k-mer on several paths of a PRG): all of them must be in the group
            bool records_ok = e.sf.y >= 1u && e.sf.y <= ((a.fuse == 2 || a.fuse == -2) ? 1u : 8u); // (fuse +-2: single-record minimizers only, for A/B runs)
            if (active && records_ok)
                for (uint32_t q = 1; q < e.sf.y; ++q) {
                    const uint32_t kq = a.rec_knode[e.sf.x + q], pq = a.rec_prg[e.sf.x + q];
                    records_ok &= pq == e.prg && (((kq & 1u) == e.strand) ? 0u : 1u) == e.rev;
                }
            const bool bad = active && (!records_ok || g != g0 || !owned || e.read == cont_read || oversized);
            if (a.dbg && a.fuse > 0) { // why entries stay behind (DRPRG_WAVE_DEBUG)
                const uint64_t b0 = __ballot(active), b1 = __ballot(active && !records_ok), b2 = __ballot(active && records_ok && g != g0),
                               b3 = __ballot(active && !owned), b4 = __ballot(active && (e.read == cont_read || oversized));
                if (lane == 0) {
                    atomicAdd(&a.dbg[0], (unsigned long long)__popcll(b0));
                    atomicAdd(&a.dbg[1], (unsigned long long)__popcll(b1));
                    atomicAdd(&a.dbg[2], (unsigned long long)__popcll(b2));
                    atomicAdd(&a.dbg[3], (unsigned long long)__popcll(b3));
                    atomicAdd(&a.dbg[4], (unsigned long long)__popcll(b4));
                }
            }
            const uint64_t bm = __ballot(bad);
            const uint64_t seg = (re >= 64u ? ~0ull : ((1ull << re) - 1ull)) & ~((1ull << rs) - 1ull);
            const bool read_bad = (bm & seg) != 0;
            // my cluster: entries [cs, ce)
            const bool chead = active && (head || (uint32_t)e.pos - pos_prev > (uint32_t)a.max_diff);
            const uint64_t cm = __ballot(chead);
            const int cs = 63 - __clzll((long long)((cm & upto) | 1ull));
            const uint64_t c_after = lane == 63 ? 0ull : (cm >> (lane + 1));
            uint32_t ce = c_after ? (uint32_t)lane + (uint32_t)__ffsll((long long)c_after) : 64u;
            if (ce > take) ce = take;
            const bool fast = active && !read_bad;
            // hits of my cluster = sum of the record counts of its entries (differences of an inclusive scan over the lanes)
            const uint32_t cnt_incl = wave_inclusive_scan(active ? e.sf.y : 0u);
            // (both shuffles by every lane: a lane that sits out a shuffle reads as 0 to the lanes that ask for its value)
            const uint32_t upto_end = (uint32_t)__shfl((int)cnt_incl, ce > 0 ? (int)ce - 1 : 0);
            const uint32_t at_prev = (uint32_t)__shfl((int)cnt_incl, cs > 0 ? cs - 1 : 0);
            const uint32_t before = cs > 0 ? at_prev : 0u;
            const bool kept = fast && (upto_end - before) > e.thr;
            if (kept) {
                for (uint32_t q = 0; q < e.sf.y; ++q) {
                    const uint32_t kq = q ? a.rec_knode[e.sf.x + q] : e.kn;
                    if (a.fuse > 0) atomicAdd(&a.covg[(kq >> 1) * 2u + e.rev], 1u);
                    else atomicSub(&a.covg[(kq >> 1) * 2u + e.rev], 1u);
                }
                if (lane == cs) {
                    if (a.fuse > 0) atomicAdd(&a.prg_reads[e.prg], 1u);
                    else atomicSub(&a.prg_reads[e.prg], 1u);
                }
            }
            fast_hits += wave_sum(kept ? e.sf.y : 0u);
            fast_clusters += (uint32_t)__popcll(__ballot(kept && lane == cs));
            if (active) my_hits += e.sf.y;
            if (fast) list[i] = (uint16_t)(e.p | 0x8000u); // done
            n_f += (uint32_t)__popcll(__ballot(active && read_bad));
            if (a.dbg && a.fuse > 0) {
                const uint64_t left = __ballot(active && read_bad);
                if (lane == 0) atomicAdd(&a.dbg[5], (unsigned long long)__popcll(left));
            }
            if (oversized) cont_read = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.read);
            c += take;
        }
        wave_lds_fence();
    }

    // ---- where this tile's records go: the workgroup's four tiles share one slice, in tile order ----
    if (lane == 0) s_nb[wave] = n_f;
    __syncthreads();
    uint32_t base = 0, wg_total = 0;
#pragma unroll
    for (int v = 0; v < SW_WAVES; ++v) {
        const uint32_t x = *LDS_U32(&s_nb[v]);
        if (v < wave) base += x;
        wg_total += x;
    }
    const size_t slice = (size_t)blockIdx.x * a.tile_cap;
    // ---- stage C2: one record per index minimizer that stage C1 left ----
    uint32_t written = 0;
    for (uint32_t i0 = 0; i0 < n_b && (!FUSE || a.fuse >= 0); i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool todo = i < n_b && (!FUSE || !(list[i] & 0x8000u));
        const uint64_t tm = __ballot(todo);
        if (todo) {
            const Entry e = decode(i);
            if (!FUSE) my_hits += e.sf.y;
            const uint32_t at = base + written + lanes_below(tm);
            if (e.pos >= (1ull << HIT_POS_BITS)) atomicOr(a.overflow, 2u);
            else if (at < a.tile_cap) {
                a.tile_info[slice + at] = ((uint64_t)e.slot << 32) | ((uint64_t)e.strand << 31) | (uint64_t)e.read;
                a.tile_pos1[slice + at] = (uint32_t)e.pos + 1;
                a.tile_rec[slice + at] = make_uint4(e.sf.x, e.sf.y, (e.strand << 31) | (((e.prg << 1) | e.rev) << 16) | e.thr, (e.kn >> 1) * 2u + e.rev);
            }
        }
        written += (uint32_t)__popcll(tm);
    }
    // ---- the workgroup's counters ----
    const uint32_t tile_hits = wave_sum(my_hits);
    if (lane == 0) {
        atomicAdd(&s_sum[0], tile_hits);
        atomicAdd(&s_sum[1], nmin);
        atomicAdd(&s_sum[2], fast_clusters);
        atomicAdd(&s_sum[3], fast_hits);
    }
    __syncthreads();
    if (wave == 0 && lane == 0 && (!FUSE || a.fuse >= 0)) {
        a.tile_count[blockIdx.x] = wg_total < a.tile_cap ? wg_total : a.tile_cap;
        a.tile_hits[blockIdx.x] = s_sum[0];
        a.tile_nmin[blockIdx.x] = s_sum[1];
        a.tile_fast[blockIdx.x] = (s_sum[3] << 16) | s_sum[2]; // hits and clusters kept by stage C1 (<= 3904 each)
        if (wg_total > a.tile_cap) atomicOr(a.overflow, 4u);
    }
}

// One tile per wave, tile = blockIdx.x * SW_WAVES + wave; the waves of a workgroup meet twice (slice offsets, counters).
//
// Tried and measured on 1.5 M tiles (10 M x 150 bp), writing the dense list from inside this kernel instead of slices +
// gather: every tile needs the number of records of all tiles before it, i.e. a chained scan (decoupled look-back) with
// waiting waves.  A ticket per tile from one device-wide counter: 18 ms (~12 ns per atomic, in series).  A ticket per
// workgroup of 8 waves x 2 tiles: 1 s (a wave's second tile starts after its first one's wait, which serialises the groups).
// A ticket per workgroup of 16 waves / 112 KB of LDS: 10 ms, 5.3 ms even without the scan (4 waves per SIMD).  No tickets,
// tile = blockIdx order, two-level scan (64 tiles per group, look-back over groups): 13 ms against 3.8 ms without the scan --
// a tile waited 48 us for its 63 group neighbours and 40 us for the previous group, four times its own 12 us.
template <int K, int W, bool FUSE, bool PACKED>
__global__ __launch_bounds__(SW_WAVES * 64) void sketch_wave_kernel(SketchArgs a, uint32_t n_tiles)
{
    __shared__ WaveLds s_lds[SW_WAVES];
    __shared__ uint32_t s_nb[SW_WAVES], s_sum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) s_sum[threadIdx.x] = 0; // (added to after the first barrier at the earliest)
    const uint32_t tile = blockIdx.x * SW_WAVES + (uint32_t)wave;
    sketch_wave_tile<K, W, FUSE, PACKED>(a, tile, tile < n_tiles, s_lds[wave], s_nb, s_sum, lane, wave);
}

hipError_t launch_sketch_wave(const SketchArgs& a, hipStream_t stream, KernelTimer timer)
{
    if (a.n_bases == 0 || !a.tile_cap || !a.prg_thr || !wave_kernel_applies(a.k, a.w)) return hipErrorInvalidValue;
    const uint32_t n_tiles = wave_n_tiles(a.n_bases);
    HIP_TRY(launch_tile_first_read(a.offsets, a.n_reads, SW_EVAL, SW_G, n_tiles, a.tile_first_read, stream));
    const dim3 g(wave_n_slices(a.n_bases)), b(SW_WAVES * 64);
    auto go = [&](auto kernel) { launch_timed(timer, kernel, g, b, 0, stream, a, n_tiles); };
    const int which = (a.w == 11 ? 0 : 4) | (a.fuse ? 2 : 0) | (a.packed ? 1 : 0);
    switch (which) {
    case 0: go(sketch_wave_kernel<15, 11, false, false>); break;
    case 1: go(sketch_wave_kernel<15, 11, false, true>); break;
    case 4: go(sketch_wave_kernel<15, 14, false, false>); break;
    case 5: go(sketch_wave_kernel<15, 14, false, true>); break;
#ifdef DRPRG_EXPERIMENTAL // stage C1 (a.fuse != 0: DRPRG_WAVE_FUSE) exists in `make EXPERIMENTAL=1` only
    case 2: go(sketch_wave_kernel<15, 11, true, false>); break;
    case 3: go(sketch_wave_kernel<15, 11, true, true>); break;
    case 6: go(sketch_wave_kernel<15, 14, true, false>); break;
    case 7: go(sketch_wave_kernel<15, 14, true, true>); break;
#endif
    default: return hipErrorInvalidValue;
    }
    HIP_TRY(hipGetLastError());
    return hipSuccess;
}

} // namespace dev
} // namespace drprg
