// sketch_filter.hip -- K1+K2(+K3) in their fast form: persistent waves + an LDS-resident Bloom prefilter, candidates that
// stay ordered by (read, position), and clusters taken straight from that order.
//
// A read minimizer can only produce a hit if its k-mer is an index k-mer (in either orientation).  So instead of
// hashing every k-mer of every read (sketch_probe.hip: ~86 VALU instructions per base, two 15-op hashes each):
//
//   this file
//   sketch_filter_kernel   every wave streams contiguous chunks of the concatenated base buffer -- a static first one, then smaller
//                          ones it draws from a counter in its workgroup's LDS (round 6: kernels.h FilterSched; the waves of a
//                          SIMD do not run at one speed) --, packs the bases to 2 bits in registers and tests the k-mer *codes*
//                          against a Bloom filter of the index k-mer codes that stays in LDS for the lifetime of the workgroup
//                          (level-0 form, k = 15: one 12-mer per four positions against a 128 KB array).  Survivors are appended
//                          in position order to the chunk's own slice (cursor in a scalar register: no atomics, no barriers;
//                          the slices in chunk order are the candidates in position order).  Three tiles of bases are in flight
//                          per wave in a register ring, the right neighbour's packed word arrives through a DPP wave shift.
//                          Level-0 form, default: the groups that pass wait in the wave's 2 KB of LDS
//                          and go through the second-stage filter -- its bits share the level-0 array -- 64 at a time,
//                          one lane per group; only surviving positions leave the kernel.
//   refine_kernel          level-0 form with DRPRG_FILTER_FORM=refine: the groups leave sketch_filter_kernel as 16-byte
//                          records; one lane per group, second-stage filter in LDS, ordered compaction per slice.
//   candidates.hip
//   verify_scan_kernel     (round 5) every workgroup scans the slice counts itself (round 6: the counts of superblocks of eight slices),
//                          takes its share of the ordered candidate list and reads the positions from the slices; then one lane per candidate, start to finish, no
//                          atomic: canonical hash from the raw bases -> exact table lookup, four slots per load (false
//                          positives end here) -> read lookup -> window-minimizer test, walked outward from the candidate
//                          (verify_lane.h).  Leaves one record per candidate and the totals per workgroup, which workgroup 0
//                          of read_cluster_kernel sums.
//   cand_scan_kernel, cand_gather_kernel, verify_count_kernel, hit_scan_kernel: the same stage as four launches over a gathered
//                          list (rounds 1-4; DRPRG_VERIFY_FORM=gather, and the generic pipeline's recount).
//   read_cluster.hip
//   read_cluster_kernel    clusters, size / overlap filters and coverage per read, out of LDS-staged chunks of the
//                          candidate list: no hit list, no sort.
//   candidates.hip, only for the reads read_cluster_kernel leaves over (long reads, many clusters)
//   recount_kernel, expand_kernel, read_inversion_kernel / read_fix_kernel: the hit list, ordered by (read, position),
//                          for the generic cluster pipeline (cluster.hip).
//
// The filter has no false negatives and every survivor is re-derived exactly from the bases, so the result is identical
// to the direct kernel (tests/test_gpu_parity.py checks both against the oracle).  Serves k <= 15, w <= 16 and indexes
// whose filter fits the LDS of a CU; everything else takes the direct kernel.
#include "filter_common.h"
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>

namespace drprg {
namespace dev {

// 16 ASCII bases -> 32 bits, 2 per base, first base in the lowest bits.  The 2-bit letter is bits 2:1 of the ASCII code
// (A 0, C 1, T 2, G 3, either case; anything else aliases one of them: it can only create a false candidate, which
// verify_count_kernel rejects from the raw bases).
__device__ inline uint32_t pack16le(const uint4& in)
{
    // v_dot4_u32_u8 with the byte weights 1, 4, 16, 64 gathers the four 2-bit fields of a dword (they sit at bits 2:1 of
    // their bytes, so the sum is twice the packed byte) at full rate; the 32 x 32 multiply that does the same is quarter rate
    constexpr uint32_t W = 0x40100401u;
    const uint32_t p0 = __builtin_amdgcn_udot4(in.x & 0x06060606u, W, 0u, false), p1 = __builtin_amdgcn_udot4(in.y & 0x06060606u, W, 0u, false);
    const uint32_t p2 = __builtin_amdgcn_udot4(in.z & 0x06060606u, W, 0u, false), p3 = __builtin_amdgcn_udot4(in.w & 0x06060606u, W, 0u, false);
    // (the doubled fields never collide -- bit 0 of each is 0 --, so the ors can be sums: two v_lshl_add_u32 where shift, shift and or3 were
    // three; spelled out, because the compiler turns the expression back into the three)
    // (Round 6 spelled the merge as two v_lshl_add_u32 in inline assembly -- one instruction fewer per word; the compiler turns the C expression
    // back into shift, shift, or3 -- and got a kernel that lost five candidates in six: the assembler statement reads the v_dot4 results, the
    // hazard recogniser does not look into it, and gfx950 wants three wait states between a dot instruction and another VALU instruction
    // that reads its result.  Not worth the two s_nop it would need.)
    return ((p0 | (p1 << 8) | (p2 << 16)) >> 1) | (p3 << 23);
}


// Bloom filter layout (built by FlatIndex, index.cpp; both orientations of every index k-mer are entered):
//   code   = little-endian 2-bit letters of the k-mer (pack16le alphabet), 2k bits
//   level 1, keyed on the first min(k,12) bases: x = code & 0xFFFFFF, h = (x * BLOOM_C1) mod 2^32,
//            word (h >> 18) & (words-1), bits 31-(h & 31), 31-((h >> 8) & 31), 31-((x >> 16) & 31)
//   level 2, keyed on the whole code: h2 = code * BLOOM_C2, word h2 >> (32-wbits), bits h2 & 31, (h2>>5) & 31, (h2>>10) & 31
//   level 0 (own array, k = 15 and small indexes), keyed on the 12-mers at offsets 0..3 of the k-mer: like level 1 with
//            BLOOM_C0; probed once per four read positions
// A level-0/1 probe costs 7 VALU + 1 LDS instruction (24-bit multiply, byte-select shifts, 3-input AND, funnel-shift
// accumulate).  Without level 0 every position pays one; with it every fourth does, level 1 runs for the ~5 % of the
// groups that pass, level 2 for the ~1 % level-1 survivors.
//
// The filter lives in dynamic LDS, which starts at LDS address 0 (the kernel has no static LDS): a masked hash is the
// address of its word, read through an address-space-3 pointer made from the integer.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ uint32_t lds_at(uint32_t byte_addr) { return *(lds_cu32*)(uintptr_t)byte_addr; }
// bit 31 of the result: all three filter bits of (h, x) are set in word
__device__ __forceinline__ uint32_t bloom_test(uint32_t word, uint32_t h, uint32_t x)
{
    uint32_t t2; // word << h[12:8]: the byte select is free, the compiler does not find it
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(t2) : "v"(h), "v"(word));
    return (word << (h & 31)) & t2 & (word << ((x >> 16) & 0xFF));
}

// Middle tier: canonical code of a 12-mer = min(code, code of its reverse complement).  v_bfrev reverses the order of the twelve
// 2-bit letters and swaps the two bits of each; the letter's low bit goes back down (>> 9), its high bit back up (>> 7) and is
// complemented there (complement of a letter = letter ^ 2).  Bits 24..31 of x are ignored.  (common.h rc12_code is the host's.)
__device__ __forceinline__ uint32_t canon12_dev(uint32_t x)
{
    const uint32_t m = x & 0xFFFFFFu;
    const uint32_t r = __builtin_bitreverse32(x);
    const uint32_t rc = ((r >> 9) & 0x555555u) | (~(r >> 7) & 0xAAAAAAu);
    return m < rc ? m : rc;
}

// SHORT_K: k < 12, the level-1 key must be masked to 2k bits.  LEVEL0: the level-0 array is present (k = 15).
// FUSED (with LEVEL0): the second stage runs in this kernel as well.  The groups that pass level 0 are staged in the wave's own
// 2 KB of LDS and, 64 at a time, put through the second-stage filter one lane per group -- dense lanes, as in refine_kernel --
// so that only surviving POSITIONS leave for global memory.  Why: the 16-byte group records of the
// two-kernel form are 104 MB of scattered writes per 10 M reads, and those writes -- not their instructions: staging them in
// LDS and copying them out coalesced once in twelve tiles changed nothing, sending them nowhere saved 65 us -- slow the
// streaming reads down.  LDS: level 0 128 KB + stage 32 KB.  A second-stage array of its own (64 KB) does
// not fit, and read from global memory (26 M four-byte loads per batch) it made this form slower than the two kernels, whether
// the words were waited for at once or a tile later; so the second-stage bits share the level-0 array (FlatIndex::bloom0f): both
// tests see a fuller array and let more through, which costs less than the records did.
// MID (with LEVEL0 and FUSED; 3 or 1 = bits per 12-mer in the level-0 array): the middle tier for indexes whose k-mers do not fit an
// LDS-resident filter.  Level 0 is keyed on the canonical 12-mer; a group that passes it is looked up in the exact bitmap of the
// canonical index 12-mers in global memory (2 MB: L2-resident; one exec-masked four-byte load per surviving group, all of a tile's
// loads in flight together), and the second stage tests the four codes of a group against a one-word Bloom filter in global memory.
// MID == 2 (round 6; with LEVEL0 and FUSED): the SMALL tier with its second stage in the L2.  Level 0 as in the all-LDS form -- plain 12-mers, three
// bits -- but the array holds level 0 ALONE (FlatIndex::bloom0: 17.5 % full for the 8d index, a random 12-mer passes at 1.2 %; with the
// second stage's six bits per code in the same array -- bloom0f, the all-LDS form -- it is 28 % full and passes 4.3 %), and a group that passes
// is tested against ONE 16-byte block of a split-block filter of the codes in global memory (FlatIndex::blkc, 256 KB for the 8d index: L2-
// resident; the block is chosen by the group's 12-mer, every index code sets one bit in each of the block's four words -- the middle tier's
// third stage, without its canonical keys and without its bitmap).  Fewer groups to stage, test and append -- 171.7 -> 140.2 M VALU and 10.0 ->
// 8.0 M LDS wave-instructions per 10 M x 150 bp -- for 11 M lane probes of the L2 (174 k gather instructions; TCC requests 12.3 -> 22.9 M), and
// the L2 answers ~267 G scattered probes a second whatever their width: 41 us of the texture path.  PACKED input has them to spare (the batch is
// a quarter of the bytes): kernel 0.240 -> 0.193 ms, step 0.387 -> 0.330.  ASCII input has not -- TA_BUSY 51 % -> 92 % of the launch, kernel
// 0.29-0.32 -> 0.33-0.35 ms on the same boxes -- so this is the default form for packed batches only (DRPRG_FILTER_STAGE2=l2|lds forces one).
// (The wait for the blocks is not what ASCII loses: a form that asked for a round's blocks behind one tile and tested them two tiles later,
// s_waitcnt vmcnt(4) instead of 0 -- checked in the ISA, nothing copied on the loop's back edge -- ran in the same 0.33-0.34 ms, and packed in
// the same 0.19: profiles/r06/schedule.txt section 9.)
// PACKED: the batch arrives as 2-bit packed words (SketchArgs::packed; the letters ARE the filter's alphabet): a lane's 32 positions
// are the two words it loads -- 8 bytes instead of 32, no pack16le (eight v_and + eight v_dot4 + the merges per tile), a quarter of
// the bytes from HBM.  Everything behind the two words is the same code, so both formats leave the same candidates.
template <bool SHORT_K, bool LEVEL0, bool FUSED = false, int MID = 0, bool PACKED = false>
__global__ __launch_bounds__(FT_THREADS) void sketch_filter_kernel(SketchArgs a, FilterWork fw)
{
    static_assert(!FUSED || LEVEL0, "the fused second stage belongs to the level-0 form");
    static_assert(MID == 0 || (LEVEL0 && FUSED), "the middle tier is a variant of the level-0 form with the second stage inside");
    extern __shared__ uint32_t s_dyn[]; // [level 0: FT_L0_WORDS] then [levels 1+2: 2^bloom_wbits words] or, FUSED, [stage: 2 KB per wave]
    // positions per lane and tile: 32 (two packed words), or -- packed input with level 0 -- 64: one 16-byte load per lane and tile, and
    // everything a tile costs once (loop control, slice bookkeeping, the ballots and prefix sums of the ordered append) is paid half as often
    constexpr int G = filter_positions_per_lane(LEVEL0, PACKED), WPOS = 63 * G, NW = G / 16, NG = G / 4;
    constexpr uint32_t L12_BASE = LEVEL0 ? FT_L0_WORDS * 4u : 0u;
    constexpr bool CANON = MID == 1 || MID == 3; // the middle tier proper: canonical 12-mers, the exact bitmap behind level 0 (MID == 2: neither)
    auto group_key = [](uint32_t x) -> uint32_t { // the 12-mer that picks a group's block of the code filter (the low 24 bits count)
        if constexpr (CANON) return canon12_dev(x);
        else return x;
    };
    (void)group_key;
    // (a wave's stage is 128 records of 16 bytes; 127 are used: the last record of the last wave's is where the chunk counter lives -- the 160 KB are full)
    constexpr uint32_t STAGE_RECORDS = 128, STAGE_CAP = STAGE_RECORDS - 1, STAGE_BASE_WORDS = FT_L0_WORDS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int k = a.k;
    const int64_t n_bases = (int64_t)a.n_bases;
    const uint32_t n_words = 1u << fw.bloom_wbits;
    const uint32_t amask = (n_words - 1) << 2; // byte offset of the level-1 word = (h >> 16) & amask
    const uint32_t amask0 = ((1u << fw.bloom0_wbits) - 1) << 2;
    const uint32_t kmask = (k < 16) ? ((1u << (2 * k)) - 1) : 0xFFFFFFFFu;
    const uint32_t kmask24 = kmask & 0xFFFFFFu;
    const int sh_w = 32 - (int)fw.bloom_wbits;
    const uint32_t sh_c = 32u - fw.midc_wbits; // (middle tier)
    (void)sh_c;
    uint32_t st_a = 0, st_b = 0, st_c = 0; // DRPRG_FT_STATS (middle tier): groups past level 0 / past the bitmap, candidate positions, per lane
    (void)st_a; (void)st_b; (void)st_c;

    // The filter arrays into LDS, 16 bytes per load and store (all of them powers of two >= 256 words, 16-byte aligned on both sides): 8 rounds
    // for the 128 KB of level 0 where word by word it was 32 -- and 8 us of every launch (round 5: step 0.511 -> 0.503 ms, packed 0.444 -> 0.437)
    auto fill = [&](uint32_t lds_word0, const uint32_t* __restrict__ from, uint32_t n) {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(from);
        uint4* dst = reinterpret_cast<uint4*>(s_dyn + lds_word0);
        for (uint32_t i = tid; i < n / 4; i += FT_THREADS) dst[i] = src[i];
    };
    if constexpr (LEVEL0) { // (FUSED: fw.bloom0 is the array that also holds the second-stage bits; always FT_L0_WORDS words: launch_sketch_filter checks)
        // all eight loads of a thread in flight before the first store: one round trip instead of eight in a row
        constexpr int ROUNDS = FT_L0_WORDS / 4 / FT_THREADS;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(fw.bloom0);
        uint4* dst = reinterpret_cast<uint4*>(s_dyn);
        uint4 v[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) v[r] = src[tid + r * FT_THREADS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) dst[tid + r * FT_THREADS] = v[r];
    }
    if (!LEVEL0) fill(L12_BASE / 4, fw.bloom, n_words);    // (the level-0 form leaves levels 1+2 to refine_kernel)
    if (tid == 0 && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_dyn != 0u) atomicOr(a.overflow, 8u);

    // ---- the schedule (FilterSched, kernels.h).  Every workgroup owns a contiguous range of the window's tiles.  Round 0: wave i of the
    // workgroup owns its chunk i, a share of 16 * tpw0 tiles; after that the wave draws tickets from the workgroup's own counter in device
    // memory, one chunk ahead of its use.  Chunk k of workgroup b = slice b * per_wg + k: the slices in order are the candidates in order. ----
    const uint32_t wv = (uint32_t)(tid >> 6);
    // the tiles that cover the bases of reads [read_begin, read_end)
    const uint64_t win_lo = a.offsets[fw.read_begin], win_hi = a.offsets[fw.read_end];
    const uint32_t t_lo = (uint32_t)(win_lo / WPOS), t_hi = (uint32_t)((win_hi + WPOS - 1) / WPOS);
    const uint32_t n_waves = gridDim.x * FT_WAVES;
    const uint32_t per_wg = fw.sched.per_wg;  // chunks = slices per workgroup
    const bool dynamic = per_wg > FT_WAVES;   // (false: one chunk per wave, as until round 5)
    // (a chunk schedule is made by the host for the tiles of the whole batch: offsets that do not span [0, n_bases) would put its chunks somewhere
    // else -- nothing is mapped then, the host reports it)
    if (dynamic && (t_lo != 0 || t_hi != fw.sched.n_tiles)) {
        if (tid == 0) atomicOr(a.overflow, 32u);
        return;
    }
    const uint32_t tiles_per_wave = fw.sched.tpw0 ? fw.sched.tpw0 : (t_hi - t_lo + n_waves - 1) / n_waves;
    // this workgroup's tiles: [wg_lo, wg_hi) -- a dynamic schedule splits the window to the tile (the schedule is made for the smaller of the
    // two sizes that gives; a workgroup's last chunk takes the odd tile), the static one gives every wave tiles_per_wave
    const uint32_t wg_lo = dynamic ? t_lo + (uint32_t)((uint64_t)blockIdx.x * (t_hi - t_lo) / gridDim.x) : (uint32_t)std::min<uint64_t>((uint64_t)t_lo + (uint64_t)blockIdx.x * FT_WAVES * tiles_per_wave, t_hi);
    const uint32_t wg_hi = dynamic ? t_lo + (uint32_t)((uint64_t)(blockIdx.x + 1) * (t_hi - t_lo) / gridDim.x) : (uint32_t)std::min<uint64_t>((uint64_t)wg_lo + (uint64_t)FT_WAVES * tiles_per_wave, t_hi);
    // The 16 waves of a workgroup do not run at one speed: a SIMD issues for its oldest wave first, so of the four waves it holds the first
    // to arrive gets what it asks for and the last what is left -- with even shares waves 0-3 of a workgroup were through after 185 us, 4-7 after
    // 227, 8-11 after 283 and 12-15 after 346 us of a 354 us launch (round 5, instrumented), the SIMDs ending the kernel on one wave each.  So the
    // shares of round 0 are uneven: fw.wave_share[c] / 256 of an even one for waves 4c .. 4c + 3, the ranges still in wave order (the slices
    // stay ordered) -- and since round 6 what round 0 leaves is handed out in chunks that get smaller towards the end, to whichever wave asks.
    if (fw.class_clock && tid == 0) atomicMax(&fw.class_clock[0], ~(unsigned long long)wall_clock64()); // (the earliest start, complemented)
    auto bound = [&](uint32_t i) -> uint32_t { // first tile of wave i of this workgroup in round 0 (i = FT_WAVES: the end of round 0)
        uint32_t cum = (i & 3u) * fw.wave_share[(i >> 2) & 3u];
        for (uint32_t c = 0; c < (i >> 2); ++c) cum += 4u * fw.wave_share[c];
        const uint64_t t = (uint64_t)wg_lo + ((uint64_t)tiles_per_wave * FT_WAVES * cum) / 4096u;
        return t < (uint64_t)wg_hi ? (uint32_t)t : wg_hi;
    };
    auto chunk_of_ticket = [&](uint32_t kt, uint32_t& b, uint32_t& e) { // kt >= FT_WAVES: a chunk of the dynamic rounds (all scalar)
        uint32_t ft = fw.sched.first_ticket[1], tl = fw.sched.first_tile[1], sz = fw.sched.size[1];
#pragma unroll
        for (int q = 2; q < FT_MAX_ROUNDS; ++q)
            if (kt >= fw.sched.first_ticket[q]) { // (0xFFFFFFFF past the last round)
                ft = fw.sched.first_ticket[q];
                tl = fw.sched.first_tile[q];
                sz = fw.sched.size[q];
            }
        b = wg_lo + tl + (kt - ft) * sz;
        e = kt + 1u == per_wg ? wg_hi : b + sz; // (the workgroup's last chunk runs to the end of its range)
    };
    // (everything of the schedule is wave-uniform, and the compiler is told so: scalar registers and scalar branches)
    uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)bound(wv));    // the current tile, in the current chunk ...
    uint32_t c_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)bound(wv + 1)); // ... which ends here
    const uint32_t slice0 = blockIdx.x * per_wg; // the workgroup's first slice
    uint32_t slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(slice0 + wv));
    // the chunk after the current one: its ticket is drawn when the prefetch first reaches past the current one -- from a counter in the workgroup's
    // LDS (fw.sched.lds_word): a round trip of ~100 ns, so nothing is reserved ahead and what a slow wave still holds when the tickets run out is
    // one chunk.  (One counter in device memory for the whole grid served ~100 tickets per microsecond -- returning atomics on one address take
    // their turn in the L2 -- and 35 k tickets made the kernel 570 us long; a counter per workgroup in device memory, drawn a chunk ahead of its
    // use to hide the 2 us, worked -- 0.333 ms -- but left the slowest waves two chunks behind the others at the end: round 6.)
    bool decoded = !dynamic, has_next = false;
    uint32_t n_k = 0, n_begin = 0, n_all = 0;
    // where chunk kt of this workgroup, tiles [b, e), keeps its candidates, and how many fit (FilterWork::slice_budget)
    auto slice_geometry = [&](uint32_t kt, uint32_t b, uint32_t e, uint32_t& base, uint32_t& cap) {
        base = blockIdx.x * fw.slice_budget + (b - wg_lo) * fw.slice_cpt + kt * fw.slice_slack;
        cap = (e - b) * fw.slice_cpt + fw.slice_slack;
    };
    uint32_t base_cur, cap_cur;
    slice_geometry(wv, tile, c_all, base_cur, cap_cur);
    base_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_cur);
    cap_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cap_cur);
    if (tid == 0) s_dyn[fw.sched.lds_word] = 0; // (before the one barrier below; a word no filter array and no wave's stage uses)
    uint32_t tiles_done = 0;
    (void)tiles_done;

    auto load16 = [&](int64_t g) -> uint4 { // 16 bases at global position g (a multiple of 16)
        if (g + 16 <= n_bases) return *reinterpret_cast<const uint4*>(a.bases + g);
        uint32_t t4[4];
        for (int q = 0; q < 4; ++q) {
            uint32_t wd = 0;
            for (int b = 0; b < 4; ++b) {
                const int64_t gg = g + q * 4 + b;
                wd |= (uint32_t)(gg < n_bases ? a.bases[gg] : (uint8_t)'N') << (8 * b);
            }
            t4[q] = wd;
        }
        return make_uint4(t4[0], t4[1], t4[2], t4[3]);
    };
    (void)load16;
    struct PairAscii { // the 32 bases of one lane in one tile
        uint4 a, b;
    };
    using Pair = typename std::conditional<PACKED, typename std::conditional<NW == 4, uint4, uint2>::type, PairAscii>::type; // (packed: the words themselves)
    const uint32_t* const words = reinterpret_cast<const uint32_t*>(a.bases);
    const int64_t n_pwords = (n_bases + 15) >> 4; // words of a packed batch
    (void)words; (void)n_pwords;
    // Tiles whose 64 x 32 bytes lie inside the buffer are loaded without any guard, so that the compiler can count the
    // loads in flight (a guarded byte path inside the loop forces s_waitcnt vmcnt(0) everywhere); the one or two
    // tiles at the very end of the buffer take the guarded path after the pipelined loop.
    // (packed: a tile is 126 words, its 64 lanes read 128: full while those lie inside the ceil(n_bases / 16) words of the batch)
    const uint32_t n_full = PACKED ? (n_pwords >= NW * 64 ? (uint32_t)((n_pwords - NW * 64) / (WPOS / 16)) + 1 : 0u)
                                   : (n_bases >= 64 * G ? (uint32_t)((n_bases - 64 * G) / WPOS) + 1 : 0u);
    uint32_t c_end = c_all < n_full ? c_all : n_full; // the current chunk's tiles that lie wholly inside the buffer end here
    auto fetch = [&](uint32_t t, Pair& p) { // unconditional; t is a full tile (the callers clamp)
        if constexpr (PACKED) {
            p = *reinterpret_cast<const Pair*>(words + (int64_t)t * (WPOS / 16) + (int64_t)lane * NW);
        } else {
            const uint8_t* g = a.bases + (int64_t)t * WPOS + (int64_t)lane * G;
            // (plain loads: non-temporal ones were measured on 10 M x 150 bp -- this kernel 382 -> 409 us, refine_kernel 45 -> 38 us
            // because the group records then survive in the L2, the step 0.720 -> 0.746 ms)
            p.a = *reinterpret_cast<const uint4*>(g);
            p.b = *reinterpret_cast<const uint4*>(g + 16);
        }
    };
    uint64_t* out = fw.raw_pos + base_cur;
    uint4* grp_out = LEVEL0 ? fw.raw_grp + base_cur : nullptr;
    (void)grp_out;
    uint32_t lane_keep = (lane == 63 || (fw.debug & 1u)) ? 0u : 0xFFFFFFFFu;
    asm volatile("" : "+v"(lane_keep)); // (a value the compiler knows nothing about: it turns a known per-lane condition back into a branch)
    uint32_t wcur = 0; // candidates in the current slice so far (wave-uniform)
    // FUSED: the groups that passed level 0 wait in the wave's 2 KB of LDS; when the next tile might not fit, they go through the
    // second stage, 64 at a time and one lane each, and the surviving positions are appended in order.  The second-stage bits live
    // in the same 128 KB array as level 0 (FlatIndex::bloom0f: there is no room for an array of their own, and reading one from
    // global memory is what made this form lose).  (A macro, not a closure or a function taking the counters by reference: those
    // kept the counters in scratch memory, whose accesses count on vmcnt like every other VMEM instruction.)
    uint4* const stage = FUSED ? reinterpret_cast<uint4*>(s_dyn + STAGE_BASE_WORDS) + (tid >> 6) * STAGE_RECORDS : nullptr;
    uint32_t lcnt = 0; // groups staged (wave-uniform)
    (void)stage;
    auto mb_below = [](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    (void)mb_below;
    // one round of the second stage: `cand2` = which of the four positions of lane's group (record r) pass, then the ordered append
#define DRPRG_STAGE2_APPEND(r, cand2_in)                                                                                              \
    do {                                                                                                                              \
        uint32_t cand2 = (cand2_in);                                                                                                  \
        /* ordered append: exclusive prefix of the per-lane counts (0..4) from three ballots */                                       \
        const uint32_t c2 = (uint32_t)__popc(cand2);                                                                                  \
        if constexpr (MID != 0) st_c += c2;                                                                                           \
        const uint64_t e0 = __ballot(c2 & 1u), e1 = __ballot(c2 & 2u), e2 = __ballot(c2 & 4u);                                        \
        if (e0 | e1 | e2) {                                                                                                           \
            uint32_t at2 = wcur + mb_below(e0) + 2u * mb_below(e1) + 4u * mb_below(e2);                                               \
            const uint64_t pos2 = ((uint64_t)(r).y << 32) | (r).x;                                                                    \
            while (cand2) {                                                                                                           \
                const int q2 = __ffs(cand2) - 1;                                                                                      \
                cand2 &= cand2 - 1;                                                                                                   \
                if (at2 < cap_cur) out[at2] = pos2 + (uint64_t)q2;                                                                    \
                ++at2;                                                                                                                \
            }                                                                                                                         \
            wcur += (uint32_t)(__popcll(e0) + 2 * __popcll(e1) + 4 * __popcll(e2));                                                   \
        }                                                                                                                             \
    } while (0)
    // middle tier: the four codes of record r against the 16-byte block `b` its 12-mer selects (one bit in each of the four words)
#define DRPRG_BLOCK_TEST(r, b, dst)                                                                                                   \
    do {                                                                                                                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                               \
            const uint32_t f = (q ? __funnelshift_r((r).z, (r).w, 2 * q) : (r).z) & kmask;                                            \
            const uint32_t h = f * BLOOM_CR;                                                                                          \
            dst |= (((b).x >> (h >> 27)) & ((b).y >> ((h >> 22) & 31)) & ((b).z >> ((h >> 17) & 31)) & ((b).w >> ((h >> 12) & 31)) & 1u) << q; \
        }                                                                                                                             \
    } while (0)
#define DRPRG_SECOND_STAGE()                                                                                                          \
    do {                                                                                                                              \
        if (lcnt) {                                                                                                                   \
            __builtin_amdgcn_wave_barrier(); /* the wave's own LDS writes come before these reads in program order */                 \
            if constexpr (MID != 0) {                                                                                                 \
                /* two rounds of 64 groups at a time, both block loads in flight before the first test */                             \
                for (uint32_t c0 = 0; c0 < lcnt; c0 += 128) {                                                                         \
                    const uint32_t ia = c0 + (uint32_t)lane, ib = ia + 64;                                                            \
                    uint4 ra = make_uint4(0, 0, 0, 0), rb = ra, ba = ra, bb = ra;                                                     \
                    if (ia < lcnt) {                                                                                                  \
                        ra = stage[ia];                                                                                               \
                        ba = reinterpret_cast<const uint4*>(fw.midc)[(uint32_t)__umul24(group_key(__funnelshift_r(ra.z, ra.w, 6)), BLOOM_C0) >> sh_c]; /* (__umul24 returns int) */ \
                    }                                                                                                                 \
                    if (ib < lcnt) {                                                                                                  \
                        rb = stage[ib];                                                                                               \
                        bb = reinterpret_cast<const uint4*>(fw.midc)[(uint32_t)__umul24(group_key(__funnelshift_r(rb.z, rb.w, 6)), BLOOM_C0) >> sh_c]; \
                    }                                                                                                                 \
                    uint32_t ca = 0, cb = 0;                                                                                          \
                    DRPRG_BLOCK_TEST(ra, ba, ca); /* (an all-zero block rejects: lanes without a record) */                           \
                    DRPRG_STAGE2_APPEND(ra, ca);                                                                                      \
                    if (c0 + 64 < lcnt) { /* wave-uniform */                                                                          \
                        DRPRG_BLOCK_TEST(rb, bb, cb);                                                                                 \
                        DRPRG_STAGE2_APPEND(rb, cb);                                                                                  \
                    }                                                                                                                 \
                }                                                                                                                     \
            } else {                                                                                                                  \
                for (uint32_t c0 = 0; c0 < lcnt; c0 += 64) {                                                                          \
                    const uint32_t ri = c0 + (uint32_t)lane;                                                                          \
                    uint32_t cs = 0;                                                                                                  \
                    uint4 r = make_uint4(0, 0, 0, 0);                                                                                 \
                    if (ri < lcnt) {                                                                                                  \
                        r = stage[ri];                                                                                                \
                        /* six bits of one word, keyed on the whole code at position q; the four words requested before the first test  \
                           (one after the other they were four LDS round trips in a row) */                                             \
                        uint32_t hq[4], h2q[4], wq[4];                                                                                \
                        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                               \
                            const uint32_t f = (q ? __funnelshift_r(r.z, r.w, 2 * q) : r.z) & kmask;                                  \
                            hq[q] = f * BLOOM_CR;                                                                                     \
                            h2q[q] = f * BLOOM_C2;                                                                                    \
                            wq[q] = lds_at((hq[q] >> 17) << 2);                                                                       \
                        }                                                                                                             \
                        __builtin_amdgcn_sched_barrier(0); /* (the scheduler otherwise pulls the tests back between the reads) */     \
                        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                 \
                            cs |= ((wq[q] >> (hq[q] & 31)) & (wq[q] >> ((hq[q] >> 5) & 31)) & (wq[q] >> ((hq[q] >> 10) & 31)) & (wq[q] >> (h2q[q] >> 27)) \
                                      & (wq[q] >> ((h2q[q] >> 22) & 31)) & (wq[q] >> ((h2q[q] >> 17) & 31)) & 1u) << q;               \
                    }                                                                                                                 \
                    DRPRG_STAGE2_APPEND(r, cs);                                                                                       \
                }                                                                                                                     \
            }                                                                                                                         \
            __builtin_amdgcn_wave_barrier();                                                                                          \
            lcnt = 0;                                                                                                                 \
        }                                                                                                                             \
    } while (0)
    auto close_slice = [&]() {
        if constexpr (FUSED) DRPRG_SECOND_STAGE(); // everything staged leaves now
        if (lane == 0) {
            const uint32_t kept = wcur < cap_cur ? wcur : cap_cur;
            (LEVEL0 && !FUSED ? fw.grp_count : fw.slice_count)[slice] = kept;
            fw.slice_base[slice] = base_cur;
            fw.slice_cap[slice] = cap_cur;
            if (wcur > cap_cur) atomicOr(a.overflow, 4u);
            if (!(LEVEL0 && !FUSED) && kept) atomicAdd(&fw.super_count[slice / FT_SUPER], kept); // (the two-kernel form: refine_kernel adds)
        }
    };

    // one tile: my 32 positions start in words wa, wb; wc (the first word of lane+1) completes the last k-mers
    auto process = [&](uint32_t t, const Pair& p) {
        uint32_t wv[NW + 1]; // my words, then the first word of lane + 1 (it completes the last k-mers)
        if constexpr (PACKED) {
            wv[0] = p.x;
            wv[1] = p.y;
            if constexpr (NW == 4) {
                wv[2] = p.z;
                wv[3] = p.w;
            }
        } else {
            wv[0] = pack16le(p.a);
            wv[1] = pack16le(p.b);
        }
        wv[NW] = __builtin_amdgcn_update_dpp(0u, wv[0], 0x130 /* wave_shl:1: lane i <- lane i+1 */, 0xF, 0xF, true); // (bound_ctrl: lane 63 reads 0, no register to clear first)
        const uint32_t wa = wv[0], wb = wv[1], wc = wv[2];
        (void)wa; (void)wb; (void)wc;
        if constexpr (LEVEL0) {
            // ---- level 0: one 12-mer per four positions (the one at 4g+3 lies inside every 15-mer starting at 4g..4g+3); group g
            // ends up in bit g of grp.  The ~2 % of the groups that pass leave the kernel as they are, with their bases: levels 1+2
            // run in refine_kernel, one lane per group (here they would run for the whole wave as often as its busiest lane
            // needs: a third of this kernel's instructions) ----
            uint32_t grp = 0, xs[NG], hs[NG], ws[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) { // all the LDS reads in flight before the first test
                const int j = 4 * g + 3;
                const uint32_t lo = wv[j >> 4], hi = wv[(j >> 4) + 1];
                xs[g] = __builtin_amdgcn_alignbit(hi, lo, 2 * (j & 15));
                if constexpr (CANON) xs[g] = canon12_dev(xs[g]);
                hs[g] = __umul24(xs[g], BLOOM_C0);
                ws[g] = lds_at((hs[g] >> 15) & amask0);
            }
#pragma unroll
            for (int g = NG - 1; g >= 0; --g)
                grp = __builtin_amdgcn_alignbit(grp, MID == 1 ? ws[g] << (hs[g] & 31) : bloom_test(ws[g], hs[g], xs[g]), 31);
            grp &= lane_keep; // (lane 63's word is only lane 62's right neighbour; as a mask: the branch cost eight scalar instructions per tile)
            if constexpr (CANON) {
                // ---- the exact bitmap of the canonical index 12-mers, only for the groups that passed level 0: one exec-masked load
                // each, all in flight before the first test (the L2 serves ~267 G such probes per second chip-wide whatever their
                // width, so every group level 0 rejects is 3.7 ps saved) ----
                st_a += (uint32_t)__popc(grp);
                uint32_t bw[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    bw[g] = 0;
                    if (grp & (1u << g)) bw[g] = fw.mid_bitmap[xs[g] >> 5];
                }
                uint32_t keep = 0;
#pragma unroll
                for (int g = 0; g < NG; ++g) keep |= ((bw[g] >> (xs[g] & 31)) & 1u) << g;
                grp &= keep;
                st_b += (uint32_t)__popc(grp);
            }
            // ---- append in (lane, group) = position order: exclusive prefix of the per-lane counts (0..NG) from four or five ballots ----
            const uint32_t cnt = (uint32_t)__popc(grp);
            const uint64_t b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u), b3 = __ballot(cnt & 8u);
            const uint64_t b4 = NG > 8 ? __ballot(cnt & 16u) : 0ull;
            if (b0 | b1 | b2 | b3 | b4) {
                auto below = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
                auto lanes_below = [&]() { return below(b0) + 2u * below(b1) + 4u * below(b2) + 8u * below(b3) + (NG > 8 ? 16u * below(b4) : 0u); };
                const uint32_t wave_total = (uint32_t)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3) + (NG > 8 ? 16 * __popcll(b4) : 0));
                const uint64_t base = (uint64_t)t * WPOS + (uint64_t)lane * G;
                auto record = [&](int g) {
                    const int q = g >> 2; // the word the group's sixteen bases start in (a select chain: no dynamic register index)
                    uint32_t lo = wv[0], hi = wv[1];
#pragma unroll
                    for (int u = 1; u < NW; ++u) {
                        lo = q == u ? wv[u] : lo;
                        hi = q == u ? wv[u + 1] : hi;
                    }
                    const uint32_t sh = (8u * (uint32_t)g) & 31u;
                    const uint64_t pos = base + 4u * (uint32_t)g;
                    return make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), __funnelshift_r(lo, hi, sh), hi >> sh);
                };
                if constexpr (FUSED) {
                    const uint32_t total = wave_total;
                    // (running only the full rounds of 64 here and keeping the remainder staged was measured: 159.8 -> 156.7 M VALU
                    // wave-instructions per 10 M reads and no change in the kernel's time, 318 against 321 us)
                    if (lcnt + total > STAGE_CAP) DRPRG_SECOND_STAGE(); // wave-uniform
                    if (total <= STAGE_CAP) {
                        uint32_t at = lcnt + lanes_below();
                        while (grp) {
                            const int g = __ffs(grp) - 1;
                            grp &= grp - 1;
                            stage[at++] = record(g);
                        }
                        lcnt += total;
                    } else { // a tile dense with index k-mers (amplicon reads): as many lanes at a time as the stage surely holds, in lane order
                        constexpr int PART_LANES = (int)STAGE_CAP / NG;
                        for (int part = 0; part < (64 + PART_LANES - 1) / PART_LANES; ++part) {
                            uint32_t gq = (lane / PART_LANES) == part ? grp : 0u;
                            const uint32_t cq = (uint32_t)__popc(gq);
                            const uint64_t q0 = __ballot(cq & 1u), q1 = __ballot(cq & 2u), q2 = __ballot(cq & 4u), q3 = __ballot(cq & 8u);
                            const uint64_t q4 = NG > 8 ? __ballot(cq & 16u) : 0ull;
                            uint32_t at = below(q0) + 2u * below(q1) + 4u * below(q2) + 8u * below(q3) + (NG > 8 ? 16u * below(q4) : 0u);
                            while (gq) {
                                const int g = __ffs(gq) - 1;
                                gq &= gq - 1;
                                stage[at++] = record(g);
                            }
                            lcnt = (uint32_t)(__popcll(q0) + 2 * __popcll(q1) + 4 * __popcll(q2) + 8 * __popcll(q3) + (NG > 8 ? 16 * __popcll(q4) : 0));
                            DRPRG_SECOND_STAGE();
                        }
                    }
                } else {
                    uint32_t at = wcur + lanes_below();
                    while (grp) {
                        const int g = __ffs(grp) - 1;
                        grp &= grp - 1;
                        if (at < cap_cur) grp_out[at] = record(g);
                        ++at;
                    }
                    wcur += wave_total;
                }
            }
            return;
        }
        uint32_t cand = 0;
        if (!(fw.debug & 1u)) {
            uint32_t c1 = 0; // level-1 survivors, position j in bit j
            {
                // ---- level 1 over my 32 positions ----
#pragma unroll
                for (int j = FT_G - 1; j >= 0; --j) {
                    const uint32_t lo = j < 16 ? wa : wb, hi = j < 16 ? wb : wc;
                    uint32_t x = (j & 15) ? __builtin_amdgcn_alignbit(hi, lo, 2 * (j & 15)) : lo; // code in the low bits, later bases above
                    if (SHORT_K) x &= kmask24;
                    const uint32_t h = __umul24(x, BLOOM_C1);
                    c1 = __builtin_amdgcn_alignbit(c1, bloom_test(lds_at(L12_BASE + ((h >> 16) & amask)), h, x), 31); // c1 = c1 << 1 | bit
                }
                if (lane == 63) c1 = 0;
            }
            // ---- level 2, only for the survivors: three more bits in a second word, keyed on the whole code ----
            while (c1) {
                const int j = __ffs(c1) - 1;
                c1 &= c1 - 1;
                const uint32_t lo = j < 16 ? wa : wb, hi = j < 16 ? wb : wc;
                const uint32_t f = __funnelshift_r(lo, hi, 2 * (j & 15)) & kmask;
                const uint32_t h2 = f * BLOOM_C2;
                const uint32_t word = lds_at(L12_BASE + ((h2 >> sh_w) << 2));
                cand |= ((word >> (h2 & 31)) & (word >> ((h2 >> 5) & 31)) & (word >> ((h2 >> 10) & 31)) & 1u) << j;
            }
        }
        // ---- append in (lane, bit) = position order; about two candidates per tile survive ----
        uint64_t m = __ballot(cand != 0);
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cand, l);
            if (lane == l) {
                const uint64_t base = (uint64_t)t * WPOS + (uint64_t)lane * G;
                uint32_t cc = c, at = wcur;
                while (cc) {
                    const int j = __ffs(cc) - 1;
                    cc &= cc - 1;
                    if (at < cap_cur) out[at] = base + (uint64_t)j;
                    ++at;
                }
            }
            wcur += (uint32_t)__popc(c);
        }
    };

    // register ring, unrolled so that no tile is copied between registers: the tile being processed plus two in
    // flight (2 KB each per wave, 16 waves: 64 KB outstanding per CU; a fourth register set made the kernel slower,
    // with the group records going to global memory and again with the second stage inside: 0.368-0.377 against 0.364-0.369 ms)
    // The tile two steps ahead of `tile` in this wave's stream: in the current chunk, or -- the ticket requested a chunk ago is waited for
    // here, and the one after it requested -- in the next; a wave without a next chunk reads its last full tile again.
    auto decode_next = [&]() {
        uint32_t drawn = 0;
        if (lane == 0) drawn = atomicAdd(&s_dyn[fw.sched.lds_word], 1u);
        const uint32_t kt = (uint32_t)FT_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        has_next = kt < per_wg;
        if (has_next) {
            n_k = kt;
            chunk_of_ticket(kt, n_begin, n_all);
            if (lane == 0 && n_begin + FT_DEPTH > (n_all < n_full ? n_all : n_full)) atomicOr(a.overflow, 16u); // (a schedule the host must not make: kernels.h FilterSched)
        }
        decoded = true;
    };
    auto ahead2 = [&]() -> uint32_t { // (the tile FT_DEPTH steps ahead)
        uint32_t ft = tile + FT_DEPTH;
        if (ft >= c_end) { // wave-uniform
            if (!decoded) decode_next();
            ft = has_next ? n_begin + (ft - c_end) : c_end - 1;
        }
        return ft;
    };
    // `tile` is done: on to the next one of the stream.  At the end of a chunk its slice is closed -- whatever is staged leaves now, the count is
    // written -- and the wave moves into the chunk it drew (if it drew none, tile stays at c_end: the caller's loop ends).  (Until late in
    // round 6 process() made the move when it met the new chunk's first tile: a compare and a branch per tile, and the scalar instructions
    // of this kernel are not free -- 2.4 cycles each where a vector instruction costs 3.7, fitted on the SQ counters of two builds.)
    auto advance = [&]() {
        ++tile;
        if constexpr (MID != 0) ++tiles_done;
        if (tile >= c_end && has_next) { // wave-uniform
            close_slice();
            slice = slice0 + n_k;
            slice_geometry(n_k, n_begin, n_all, base_cur, cap_cur);
            out = fw.raw_pos + base_cur;
            if (LEVEL0) grp_out = fw.raw_grp + base_cur;
            wcur = 0;
            tile = n_begin;
            c_all = n_all;
            c_end = n_all < n_full ? n_all : n_full;
            has_next = false;
            decoded = false;
        }
    };
    Pair r0 {}, r1 {}, r2 {};
    const bool pipelined = tile < c_end; // wave-uniform
    if (pipelined) {
        fetch(tile, r0);
        fetch(tile + 1 < c_end ? tile + 1 : tile, r1); // (a dynamic schedule's chunks hold FT_DEPTH full tiles at least)
    }
#if DRPRG_FT_DEPTH == 3
    Pair r3 {};
    if (pipelined) fetch(tile + 2 < c_end ? tile + 2 : tile, r2);
#endif
    __syncthreads(); // Bloom filter in place; the only barrier
    // (middle tier: a tile's bitmap probes are waited for inside process() together with every load the wave has issued, the freshly
    // requested tile included; requesting that tile BEHIND process() instead was measured and changed nothing -- 522 / 534 / 1254 us
    // against 520 / 530 / 1234 on the dense, 2-fold and 8-fold indexes: the other three waves of the SIMD cover the wait)
    if (pipelined)
        for (;;) {
#if DRPRG_FT_DEPTH == 3
            fetch(ahead2(), r3);
            process(tile, r0);
            advance();
            if (tile >= c_end) break;
            fetch(ahead2(), r0);
            process(tile, r1);
            advance();
            if (tile >= c_end) break;
            fetch(ahead2(), r1);
            process(tile, r2);
            advance();
            if (tile >= c_end) break;
            fetch(ahead2(), r2);
            process(tile, r3);
            advance();
            if (tile >= c_end) break;
#else
            fetch(ahead2(), r2);
            process(tile, r0);
            advance();
            if (tile >= c_end) break;
            fetch(ahead2(), r0);
            process(tile, r1);
            advance();
            if (tile >= c_end) break;
            fetch(ahead2(), r1);
            process(tile, r2);
            advance();
            if (tile >= c_end) break;
#endif
        }
    for (; tile < c_all; ++tile) { // the end of the buffer (the last chunk of the window only), guarded loads
        const int64_t g = (int64_t)tile * WPOS + (int64_t)lane * G;
        Pair p;
        if constexpr (PACKED) {
            const int64_t wi = g >> 4; // (words past the end read as 'A's: candidates there fail verify_count_kernel's bounds)
            p.x = wi < n_pwords ? words[wi] : 0u;
            p.y = wi + 1 < n_pwords ? words[wi + 1] : 0u;
            if constexpr (NW == 4) {
                p.z = wi + 2 < n_pwords ? words[wi + 2] : 0u;
                p.w = wi + 3 < n_pwords ? words[wi + 3] : 0u;
            }
        } else {
            p.a = load16(g);
            p.b = load16(g + 16);
        }
        process(tile, p);
        if constexpr (MID != 0) ++tiles_done;
    }
    close_slice();
    if (fw.class_clock && (tid & 255) == 0) atomicMax(&fw.class_clock[1 + (tid >> 8)], (unsigned long long)wall_clock64()); // (waves 4c: when class c was through)
    if constexpr (MID != 0)
        if (fw.stat) {
            atomicAdd(&fw.stat[1], (unsigned long long)st_a);
            atomicAdd(&fw.stat[2], (unsigned long long)st_b);
            atomicAdd(&fw.stat[3], (unsigned long long)st_c);
            if (lane == 0) atomicAdd(&fw.stat[0], (unsigned long long)tiles_done * 63ull * (G / 4));
        }
}
#undef DRPRG_SECOND_STAGE
#undef DRPRG_BLOCK_TEST
#undef DRPRG_STAGE2_APPEND

#ifdef DRPRG_EXPERIMENTAL
// Second stage of the filter for the groups that passed level 0 (level-0 form of sketch_filter_kernel): one lane per
// group tests its four k-mer codes against a 64 KB LDS-resident filter (four bits per code, < 15 % full); every wave
// works through whole slices and compacts the surviving positions, in order, into the slice of raw_pos that
// cand_scan_kernel / cand_gather_kernel expect.
constexpr int RF_THREADS = 1024;
__global__ __launch_bounds__(RF_THREADS) void refine_kernel(SketchArgs a, FilterWork fw)
{
    extern __shared__ uint32_t s_bloom[]; // the second-stage filter: 2^BLOOMR_WBITS words
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t kmask = (1u << (2 * a.k)) - 1; // k = 15
    for (uint32_t i = tid; i < (1u << BLOOMR_WBITS); i += RF_THREADS) s_bloom[i] = fw.bloomr[i];
    __syncthreads(); // the only barrier: from here on every wave works through its own slices
    auto below = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    const uint32_t n_waves = gridDim.x * (RF_THREADS / 64);
    for (uint32_t s = blockIdx.x * (RF_THREADS / 64) + (uint32_t)(tid >> 6); s < fw.n_slices; s += n_waves) {
        const uint32_t n = fw.grp_count[s], cap = fw.slice_cap[s];
        const uint4* __restrict__ in = fw.raw_grp + fw.slice_base[s];
        uint64_t* __restrict__ out = fw.raw_pos + fw.slice_base[s];
        uint32_t written = 0; // wave-uniform
        uint4 nxt = (uint32_t)lane < n ? in[lane] : make_uint4(0, 0, 0, 0);
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const uint4 r = nxt;
            const uint32_t i = c0 + (uint32_t)lane;
            if (i + 64 < n) nxt = in[i + 64]; // in flight while this chunk is tested
            uint32_t cand = 0;
            if (i < n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { // four bits of one word, keyed on the whole code of the k-mer at position q
                    const uint32_t f = (q ? __funnelshift_r(r.z, r.w, 2 * q) : r.z) & kmask;
                    const uint32_t h = f * BLOOM_CR;
                    const uint32_t word = s_bloom[h >> (32 - BLOOMR_WBITS)];
                    cand |= ((word >> (h & 31)) & (word >> ((h >> 5) & 31)) & (word >> ((h >> 10) & 31)) & (word >> ((h >> 15) & 31)) & 1u) << q;
                }
            }
            // ordered append: exclusive prefix of the per-lane counts (0..4) from three ballots
            const uint32_t cnt = (uint32_t)__popc(cand);
            const uint64_t b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
            if (b0 | b1 | b2) {
                uint32_t at = written + below(b0) + 2u * below(b1) + 4u * below(b2);
                const uint64_t pos = ((uint64_t)r.y << 32) | r.x;
                while (cand) {
                    const int q = __ffs(cand) - 1;
                    cand &= cand - 1;
                    if (at < cap) out[at] = pos + (uint64_t)q;
                    ++at;
                }
                written += (uint32_t)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2));
            }
        }
        if (lane == 0) {
            const uint32_t kept = written < cap ? written : cap;
            fw.slice_count[s] = kept;
            if (written > cap) atomicOr(a.overflow, 4u);
            if (kept) atomicAdd(&fw.super_count[s / FT_SUPER], kept);
        }
    }
}
#endif // DRPRG_EXPERIMENTAL

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// DRPRG_FILTER_FORM=refine with the library of `make EXPERIMENTAL=1`: the level-0 survivors leave sketch_filter_kernel as 16-byte group
// records (FilterBuffers::raw_grp) for refine_kernel.  Read at every launch; the host allocates raw_grp only then (16 bytes per
// candidate slot that the default sequence never touches).
bool group_records_requested()
{
#ifdef DRPRG_EXPERIMENTAL
    const char* form = std::getenv("DRPRG_FILTER_FORM");
    return form && std::string(form) == "refine";
#else
    return false;
#endif
}

uint32_t filter_n_tiles(uint64_t n_bases, int positions_per_lane)
{
    const uint64_t wpos = 63ull * (uint64_t)positions_per_lane;
    return (uint32_t)((n_bases + wpos - 1) / wpos);
}

uint32_t filter_grid(bool level0, int n_cus, uint32_t n_tiles)
{
    // persistent grid: the workgroups that stay resident (1024 threads each; 64 KB of LDS: two per CU, 128 KB with
    // level 0: one per CU), never more waves than there are wave tiles
    uint32_t grid = (uint32_t)n_cus * (level0 ? 1 : 2);
    const uint32_t need = (n_tiles + FT_WAVES - 1) / FT_WAVES;
    if (grid > need) grid = need;
    if (grid > (uint32_t)(MAX_SLICES / FT_WAVES)) grid = MAX_SLICES / FT_WAVES;
    if (const char* e = std::getenv("DRPRG_FT_GRID")) { // (tests: few workgroups, so that a small batch gives every wave a chunk schedule)
        const int cap = std::atoi(e);
        if (cap >= 1 && grid > (uint32_t)cap) grid = (uint32_t)cap;
    }
    return grid ? grid : 1;
}

// FilterBuffers::small: [slice counts: MAX_CHUNKS][superblock counts: MAX_SLICES][slice starts: MAX_CHUNKS][slice rooms: MAX_CHUNKS][prefix of the
// slices: MAX_CHUNKS + 4 (gathered form only; the word behind the last slice's is the batch's candidate total)][group counts: MAX_CHUNKS (two-kernel
// form)][per-workgroup totals: 4 x MAX_EX_WG].  The superblock counts are zero between two sequences (counters_home_kernel).
constexpr size_t FT_SMALL_HEAD = (size_t)MAX_CHUNKS * 3 + MAX_SLICES;
size_t filter_small_words() { return FT_SMALL_HEAD + (size_t)MAX_CHUNKS * 2 + 4 + 4 * (size_t)MAX_EX_WG; }
uint32_t* filter_super_counts(uint32_t* small) { return small + MAX_CHUNKS; }
uint32_t filter_super_words() { return (uint32_t)MAX_SLICES; }

FilterSched make_filter_sched(uint32_t n_tiles, uint32_t n_wg, bool window_known, const uint32_t share[4])
{
    FilterSched s {};
    for (int r = 0; r < FT_MAX_ROUNDS; ++r) s.first_ticket[r] = 0xFFFFFFFFu;
    s.n_rounds = 1;
    s.per_wg = FT_WAVES;
    s.n_tiles = n_tiles;
    s.tpw0 = window_known ? (n_tiles + n_wg * FT_WAVES - 1) / (n_wg * FT_WAVES) : 0u;
    // DRPRG_FT_SCHED (read at every launch: tests switch it): "static", or "f,d,m[,a]" = round 0's part of the tiles in 1/256, the divisor of
    // the dynamic rounds x 16 (a round hands every wave 16 / d of an even share of what is left), the smallest chunk in tiles, and the
    // fewest tiles per wave a batch must have for a dynamic schedule at all (64: below that the static split is as good, and cheaper)
    struct Knobs {
        bool is_static = false;
        uint32_t f = 180, d = 32, m = 4, min_avg = 64;
    } knobs;
    if (const char* e = std::getenv("DRPRG_FT_SCHED")) {
        if (std::string(e) == "static") knobs.is_static = true;
        else {
            unsigned a = 0, b = 0, c = 0, g = 64;
            const int got = std::sscanf(e, "%u,%u,%u,%u", &a, &b, &c, &g);
            if (got >= 3 && a >= 1 && a <= 250 && b >= 17 && b <= 1024 && c >= 4 && c <= 4096 && g >= 8) {
                knobs.f = a;
                knobs.d = b;
                knobs.m = c;
                knobs.min_avg = g;
            }
        }
    }
    uint32_t min_share = share[0];
    for (int c = 1; c < 4; ++c) min_share = std::min(min_share, share[c]);
    // the schedule of ONE workgroup, for the smaller of the two sizes an even split of the window gives (the kernel's workgroups find their
    // own ranges; the last chunk of each takes the odd tile)
    const uint64_t wg_tiles = n_tiles / n_wg, avg = wg_tiles / FT_WAVES;
    const uint32_t tpw0 = (uint32_t)(avg * knobs.f / 256);
    // every chunk of a dynamic schedule must hold two tiles that lie wholly inside the buffer (the last two tiles of a batch may not): the
    // smallest share of round 0 -- the class split rounds down by up to a tile -- and the smallest chunk say whether this batch can have one
    if (!window_known || knobs.is_static || avg < knobs.min_avg || (uint64_t)tpw0 * min_share / 256 < FT_DEPTH + 2) return s;
    const uint32_t max_per_wg = (uint32_t)MAX_CHUNKS / n_wg;
    uint64_t done = (uint64_t)FT_WAVES * tpw0, rest = wg_tiles - done;
    if (rest < FT_DEPTH + 2) return s; // (the last chunk -- the whole dynamic part here -- must hold FT_DEPTH whole tiles: the batch's last two may be partial)
    uint32_t tk = FT_WAVES;
    int r = 1;
    for (; r < FT_MAX_ROUNDS - 1 && tk + 2 * FT_WAVES <= max_per_wg; ++r) {
        const uint32_t sz = (uint32_t)(rest * 16 / ((uint64_t)FT_WAVES * knobs.d));
        if (sz < 2 * knobs.m) break;
        s.first_ticket[r] = tk;
        s.first_tile[r] = (uint32_t)done;
        s.size[r] = sz;
        done += (uint64_t)FT_WAVES * sz;
        rest -= (uint64_t)FT_WAVES * sz;
        tk += FT_WAVES;
    }
    // the last round: chunks of the smallest size, or larger ones if the slices would not suffice; its last chunk takes the remainder
    const uint32_t left = max_per_wg - tk;
    const uint32_t sz = (uint32_t)std::max<uint64_t>(knobs.m, (rest + left - 1) / left);
    const uint32_t cnt = (uint32_t)std::max<uint64_t>(1, (rest - (FT_DEPTH - 2)) / sz); // (the last chunk: sz + FT_DEPTH - 2 tiles at least)
    s.first_ticket[r] = tk;
    s.first_tile[r] = (uint32_t)done;
    s.size[r] = sz;
    s.n_rounds = (uint32_t)r + 1;
    s.per_wg = tk + cnt;
    s.tpw0 = tpw0;
    return s;
}

void init_candidate_work(FilterWork& fw, const FilterBuffers& b, int n_cus)
{
    fw.cand_gp = b.cand_gp;
    fw.cand_info = b.cand_info;
    fw.cand_pos1 = b.cand_pos1;
    fw.cand_rec = b.cand_rec;
    fw.wg_hits = b.small + FT_SMALL_HEAD + 2 * MAX_CHUNKS + 4;
    fw.wg_nmin = fw.wg_hits + MAX_EX_WG;
    fw.wg_maxlen = fw.wg_nmin + MAX_EX_WG;
    fw.wg_base = fw.wg_maxlen + MAX_EX_WG;
    fw.ex_grid = std::min<uint32_t>((uint32_t)n_cus * 8, MAX_EX_WG);
    fw.verify_grid = fw.ex_grid;
    fw.max_len = b.max_len;
}

hipError_t launch_sketch_filter(const SketchArgs& a, uint32_t read_begin, uint32_t read_end, const BloomTables& bt, int n_cus,
    const FilterBuffers& b, const ReadClusterArgs& rc, FilterWork& fw, hipStream_t stream, KernelTimer timer)
{
    fw = FilterWork {};
    fw.read_begin = read_begin;
    fw.read_end = read_end;
    if (a.n_bases == 0) return hipSuccess;
    const bool mid = bt.mid_bitmap != nullptr; // middle tier: level 0 (canonical 12-mers) in LDS, bitmap + code filter in global memory
    if (mid && (a.k != 15 || !bt.mid0 || !bt.midc || bt.midc_wbits < 1 || bt.midc_wbits > MID_C_MAX_WBITS || (bt.mid0_bits != 1 && bt.mid0_bits != 3)))
        return hipErrorInvalidValue;
    if (!mid && (1u << bt.bloom_wbits) > (uint32_t)FT_BLOOM_WORDS) return hipErrorInvalidValue;
    if (!mid && bt.bloom0 && (1u << bt.bloom0_wbits) != (uint32_t)FT_L0_WORDS) return hipErrorInvalidValue;
    if (!mid && bt.bloom0 && (!bt.bloomr || !bt.bloom0f)) return hipErrorInvalidValue;
    if (const char* dbg = std::getenv("DRPRG_FT_DEBUG")) fw.debug = (uint32_t)std::atoi(dbg); // 1 no filter test, 4 no level 0, 8 no read_cluster_kernel
    const bool level0 = mid || (bt.bloom0 != nullptr && a.k == 15 && ((size_t)4 << bt.bloom_wbits) + (size_t)FT_L0_WORDS * 4 <= 160 * 1024
        && !(fw.debug & 4u));
    // the second stage inside the streaming kernel (default; its bits share the level-0 array) or as refine_kernel behind it
    // (DRPRG_FILTER_FORM=refine): 0.64 against 0.69 ms per 10 M reads, DESIGN.md section 6
    const bool fused = mid || (level0 && !group_records_requested()); // (the two-kernel form is part of `make EXPERIMENTAL=1` only)
    // Small tier: the second stage against a block filter in the L2 (sketch_filter_kernel<.., MID = 2>: the level-0 array then holds level 0 alone and
    // lets fewer groups through -- 11 M per 10 M x 150 bp) for packed batches, round 2-5's all-LDS form for ASCII ones; DRPRG_FILTER_STAGE2=l2|lds asks for one of
    // the two whatever the format (read at every launch: A/B runs, tests).  Both leave the same candidates behind verify: the tests map with both.
    const char* const stage2 = std::getenv("DRPRG_FILTER_STAGE2");
    const bool want_l2 = stage2 && *stage2 ? std::string(stage2) == "l2" : a.packed != 0; // (the kernel's header comment: packed batches have the L2 probes to spare)
    const bool blk = !mid && level0 && fused && want_l2 && bt.blkc != nullptr && bt.blkc_wbits >= 1 && bt.blkc_wbits <= MID_C_MAX_WBITS;
    if (level0 && !fused && !b.raw_grp) return hipErrorInvalidValue; // (the caller allocates the group records when group_records_requested())
    fw.bloom = bt.bloom;
    fw.bloom_wbits = bt.bloom_wbits;
    fw.bloomr = bt.bloomr;
    fw.bloom0 = mid ? bt.mid0 : (level0 ? (fused && !blk ? bt.bloom0f : bt.bloom0) : nullptr);
    fw.bloom0_wbits = mid ? 15 : (level0 ? bt.bloom0_wbits : 0);
    fw.mid_bitmap = bt.mid_bitmap;
    fw.midc = blk ? bt.blkc : bt.midc;
    fw.midc_wbits = blk ? bt.blkc_wbits : bt.midc_wbits;
    fw.stat = mid ? b.stat : nullptr;
    const uint32_t grid = filter_grid(level0, n_cus, filter_n_tiles(a.n_bases, filter_positions_per_lane(level0, a.packed != 0)));
    {   // The shares of the four wave classes of a workgroup (sketch_filter_kernel: a SIMD issues for its oldest wave first).  Measured where the
        // classes of the level-0 forms (one workgroup per CU, four waves per SIMD) end with even shares -- 8d index 208 / 243 / 291 / 351 us of a 366 us
        // launch, packed 139 / 184 / 237 / 292 of 309 -- and set so that they end together there: 306 / 288 / 286 / 304 of 320 us, packed 219 / 226 /
        // 240 / 257 of 270.  The other workloads of the level-0 forms gain less from the same shares (middle tier, dense 8d genes 501 -> 479 us,
        // 8-fold index 1252 -> 1219; 4 kb reads 1966 -> 1801) and none loses; the forms with two workgroups per CU keep even shares.
        // DRPRG_FT_SHARE=a,b,c,d (any scale): measurements.  Any shares give the same candidates: the ranges stay in wave order.
        // (read at every launch: the tests map one batch with several)
        const char* const share_env = std::getenv("DRPRG_FT_SHARE");
        const bool from_env = share_env != nullptr;
        std::array<uint32_t, 4> env_share { 256, 256, 256, 256 };
        if (share_env) {
            double v[4] = { 1, 1, 1, 1 };
            if (std::sscanf(share_env, "%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0) {
                const double sum = v[0] + v[1] + v[2] + v[3];
                uint32_t acc = 0;
                for (int c = 0; c < 3; ++c) acc += env_share[c] = (uint32_t)(1024.0 * v[c] / sum + 0.5);
                env_share[3] = 1024u - acc;
            }
        }
        // (the middle tier's waves wait for the L2 more and for each other less: its classes end at 1 : 1.13 : 1.30 : 1.49 with even shares)
        static const uint32_t even[4] = { 256, 256, 256, 256 }, ascii_l0[4] = { 397, 294, 200, 133 }, packed_l0[4] = { 422, 292, 184, 126 }, mid_l0[4] = { 356, 292, 220, 156 };
        const uint32_t* share = from_env ? env_share.data() : !level0 ? even : b.wave_share ? b.wave_share : mid ? mid_l0 : a.packed ? packed_l0 : ascii_l0;
        fw.class_clock = level0 && !from_env ? b.class_clock : nullptr;
        for (int c = 0; c < 4; ++c) fw.wave_share[c] = share[c];
    }
    {   // the chunk schedule: dynamic when this sequence covers the whole batch (the host then knows the tile numbers) and the batch is large enough
        const bool whole = read_begin == 0 && read_end == a.n_reads;
        fw.sched = make_filter_sched(filter_n_tiles(a.n_bases, filter_positions_per_lane(level0, a.packed != 0)), grid, whole, fw.wave_share);
    }
    fw.n_slices = grid * fw.sched.per_wg;
    {   // the slices' geometry (FilterWork::slice_budget): a floor of up to 256 entries per slice -- a quarter of the workgroup's budget at most --
        // and the rest by the tile.  (A workgroup's range: an even split to the tile, or -- static schedule -- 16 waves x tiles per wave.)
        const uint32_t n_tiles = filter_n_tiles(a.n_bases, filter_positions_per_lane(level0, a.packed != 0));
        const uint64_t budget = std::min<uint64_t>(b.raw_capacity, 0x7FFFFFFFull) / grid;
        const uint64_t wg_tiles = fw.sched.per_wg > (uint32_t)FT_WAVES ? (n_tiles + grid - 1) / grid : (uint64_t)FT_WAVES * ((n_tiles + grid * FT_WAVES - 1) / (grid * FT_WAVES));
        fw.slice_budget = (uint32_t)budget;
        fw.slice_slack = (uint32_t)std::min<uint64_t>(256, budget / (4ull * fw.sched.per_wg));
        fw.slice_cpt = (uint32_t)((budget - (uint64_t)fw.slice_slack * fw.sched.per_wg) / std::max<uint64_t>(wg_tiles, 1));
    }
    fw.raw_pos = b.raw_pos;
    fw.cand_info = b.cand_info;
    fw.cand_pos1 = b.cand_pos1;
    fw.cand_rec = b.cand_rec;
    init_candidate_work(fw, b, n_cus);
    fw.slice_count = b.small;
    fw.super_count = filter_super_counts(b.small);
    fw.slice_base = b.small + MAX_CHUNKS + MAX_SLICES;
    fw.slice_cap = fw.slice_base + MAX_CHUNKS;
    fw.cand_prefix = b.small + FT_SMALL_HEAD;
    fw.cand_total = fw.cand_prefix + fw.n_slices;
    fw.grp_count = b.small + FT_SMALL_HEAD + MAX_CHUNKS + 4;
    fw.raw_grp = b.raw_grp;
    {
        using Kernel = void (*)(SketchArgs, FilterWork);
        const int which = mid ? (bt.mid0_bits == 3 ? 4 : 5) : blk ? 6 : level0 ? (fused ? 3 : 2) : (a.k < 12 ? 1 : 0);
        const Kernel ascii = which == 6 ? &sketch_filter_kernel<false, true, true, 2>
            : which == 5                ? &sketch_filter_kernel<false, true, true, 1>
            : which == 4                ? &sketch_filter_kernel<false, true, true, 3>
            : which == 3                ? &sketch_filter_kernel<false, true, true>
            : which == 2                ? &sketch_filter_kernel<false, true>
            : which == 1                ? &sketch_filter_kernel<true, false>
                                        : &sketch_filter_kernel<false, false>;
        const Kernel packed = which == 6 ? &sketch_filter_kernel<false, true, true, 2, true>
            : which == 5                 ? &sketch_filter_kernel<false, true, true, 1, true>
            : which == 4                 ? &sketch_filter_kernel<false, true, true, 3, true>
            : which == 3                 ? &sketch_filter_kernel<false, true, true, 0, true>
            : which == 2                 ? &sketch_filter_kernel<false, true, false, 0, true>
            : which == 1                 ? &sketch_filter_kernel<true, false, false, 0, true>
                                         : &sketch_filter_kernel<false, false, false, 0, true>;
        const Kernel kernel = a.packed ? packed : ascii;
        // (+ the chunk counter: 16 bytes of their own, or -- the level-0 form with the second stage inside fills the 160 KB -- the last record of the last wave's stage)
        const size_t dyn = level0 ? (size_t)FT_L0_WORDS * 4 + (fused ? (size_t)FT_WAVES * 2048 : 16) : ((size_t)4 << bt.bloom_wbits) + 16;
        fw.sched.lds_word = (uint32_t)(dyn / 4 - 4);
        static size_t configured[14][MAX_HIP_DEVICES] = {};
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), dyn, configured[which + (a.packed ? 7 : 0)]));
        launch_timed(timer, kernel, dim3(grid), dim3(FT_THREADS), dyn, stream, a, fw);
    }
    HIP_TRY(hipGetLastError());
#ifdef DRPRG_EXPERIMENTAL
    if (level0 && !fused) {
        static size_t refine_configured[MAX_HIP_DEVICES] = {};
        const size_t dyn = (size_t)4 << BLOOMR_WBITS;
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&refine_kernel), dyn, refine_configured));
        hipLaunchKernelGGL(refine_kernel, dim3(std::min<uint32_t>((fw.n_slices + RF_THREADS / 64 - 1) / (RF_THREADS / 64), (uint32_t)n_cus * 2)), dim3(RF_THREADS), dyn, stream, a, fw);
        HIP_TRY(hipGetLastError());
    }
#endif
    const bool skip_rc = (fw.debug & 8u) != 0; // debug 8: every read with a hit goes the generic way
    HIP_TRY(launch_candidate_stage(a, fw, rc, n_cus, stream, skip_rc));
    ReadClusterArgs rct = rc;
    if (!skip_rc) { // the batch totals come out of read_cluster_kernel's workgroup 0
        rct.wg_hits = fw.wg_hits;
        rct.wg_nmin = fw.wg_nmin;
        rct.wg_maxlen = fw.wg_maxlen;
        rct.n_wg = fw.verify_grid;
        rct.tot_hits = a.n_hits;
        rct.tot_minimizers = a.n_minimizers;
        rct.tot_max_len = fw.max_len;
        rct.overflow_word = a.overflow;
    }
    return launch_read_cluster(a, fw, rct, n_cus, skip_rc, stream);
}

} // namespace dev
} // namespace drprg
