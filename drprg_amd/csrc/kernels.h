// kernels.h -- argument blocks and launch wrappers of kernels.hip (host side sees only hip_runtime_api).
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime_api.h>

#if defined(__HIPCC__)
#define DRPRG_HD __host__ __device__
#else
#define DRPRG_HD
#endif

namespace drprg {
namespace dev {

// hit sort key: [read:28][prg:12][rev:1][pos:23]; forward hits (rev=0) sort first
constexpr int HIT_POS_BITS = 23;
constexpr int HIT_PRG_BITS = 12;
constexpr int HIT_READ_BITS = 28;
constexpr uint64_t HIT_POS_MASK = (1ull << HIT_POS_BITS) - 1;
constexpr uint32_t MAX_BATCH_READS = 1u << HIT_READ_BITS;
constexpr uint32_t MAX_PRGS = 1u << HIT_PRG_BITS;

DRPRG_HD inline uint64_t pack_hit_key(uint32_t read, uint32_t prg, uint32_t rev, uint32_t pos)
{
    return ((uint64_t)read << (HIT_PRG_BITS + 1 + HIT_POS_BITS)) | ((uint64_t)prg << (1 + HIT_POS_BITS))
        | ((uint64_t)rev << HIT_POS_BITS) | (uint64_t)pos;
}
DRPRG_HD inline uint32_t hit_read(uint64_t k) { return (uint32_t)(k >> (HIT_PRG_BITS + 1 + HIT_POS_BITS)); }
DRPRG_HD inline uint32_t hit_prg(uint64_t k) { return (uint32_t)(k >> (1 + HIT_POS_BITS)) & (MAX_PRGS - 1); }
DRPRG_HD inline uint32_t hit_rev(uint64_t k) { return (uint32_t)(k >> HIT_POS_BITS) & 1u; }

// Bloom tier in front of the hash table (direct kernel, large indexes): word and two bits from one multiplicative hash
DRPRG_HD inline uint32_t pbloom_mix(uint64_t key) { return ((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u; }
DRPRG_HD inline uint32_t pbloom_word(uint32_t m, uint32_t wbits) { return m >> (32 - wbits); }
DRPRG_HD inline uint32_t pbloom_bits(uint32_t m) { return (1u << (m & 31)) | (1u << ((m >> 5) & 31)); }

struct SketchArgs {
    const uint8_t* bases;    // 16-byte aligned.  packed != 0: the same pointer is u32 words[ceil(n_bases / 16)], 2 bits per base
    const uint64_t* offsets; // n_reads + 1 (in bases, whatever the format)
    uint64_t n_bases;
    uint32_t n_reads;
    int w, k, halo;
    // 2-bit packed reads (SURVEY.md section 8f NEXT-4; include/drprg_hip.h "packed reads"): base i of the batch is bits
    // [2 (i & 15) + 1 : 2 (i & 15)] of word i >> 4, letter = bits 2:1 of its ASCII code (A 0, C 1, T 2, G 3 -- the alphabet of the
    // filter's k-mer codes, sketch_filter.hip pack16le -- for ANY byte, so that both formats give the filter the same codes);
    // npos: the ascending positions of the bases that are not one of ACGTacgt (they poison every k-mer that holds them, exactly
    // as in the ASCII form).  The kernels of the filtered sequence and sketch_wave_kernel read this form; every other consumer of
    // bases gets an ASCII copy made on the device (launch_unpack).
    int packed;
    const uint64_t* npos;
    uint64_t n_npos;
    const uint16_t* nbits; // packed batches in sketch_wave_kernel: bit j of word i = base 16 i + j is in npos (null when n_npos == 0):
                           // launch_mark_npos sets them before the kernel (the list itself would cost a search per tile)
    // index
    const void* slot_key; // u32[2^bits] (k <= 15) or u64[2^bits]
    const uint2* slot_rec; // {record offset, record count}; count 0 = empty slot
    const uint4* slot_first; // {record offset, record count, rec_knode of the first record, its prg | min path length of that prg << 12}:
                             // everything a candidate record needs of a slot in one load instead of three dependent ones
    uint32_t table_bits;
    const uint32_t* rec_knode; // (global k-mer node << 1) | strand
    const uint16_t* rec_prg;
    uint32_t* tile_first_read; // scratch, one entry per tile (filled by launch_sketch_probe)
    // outputs
    uint64_t* hit_key;
    uint32_t* hit_val;
    uint64_t hit_capacity;
    unsigned long long* n_hits;
    unsigned long long* n_minimizers;
    const uint32_t* pbloom; // direct kernel: Bloom tier in front of a table that does not fit L2 (nullptr: none)
    uint32_t pbloom_wbits;
    uint32_t* overflow; // bit 0: hit buffer too small, bit 1: read longer than 2^HIT_POS_BITS, bit 2: candidate slice
                        // too small, bit 3: dynamic LDS does not start at address 0 (sketch_filter_kernel), bit 4: a chunk of its schedule holds too
                        // few whole tiles, bit 5: the batch's offsets do not span [0, n_bases) (its chunk schedule was made for all of its tiles)
    // candidate form of the direct kernel (tile_cap != 0): instead of hits, every tile leaves the records
    // read_cluster_kernel wants (same layout as FilterWork::cand_info / cand_pos1 / cand_rec), in position order, in its own
    // slice of tile_cap entries; tile_count[t] = minimizers found, tile_hits[t] = their hits, tile_nmin[t] = all minimizers of the tile
    uint32_t tile_cap;
    uint64_t* tile_info;
    uint32_t* tile_pos1;
    uint4* tile_rec;
    uint32_t *tile_count, *tile_hits, *tile_nmin;
    const uint32_t* prg_min_path_len; // for the size threshold stored in the records
    const uint32_t* prg_thr;          // per PRG: floor(shortest k-mer path * fraction) (sketch_wave_kernel)
    // in-kernel clustering of the reads that lie inside one tile (sketch_wave_kernel stage C1): 0 off, 1 on, -1 undo
    int fuse;
    int max_diff;
    uint32_t* covg;      // the batch's accumulators (stage C1 adds to them directly)
    uint32_t* prg_reads;
    uint32_t* tile_fast; // per slice: hits kept << 16 | clusters kept by stage C1
    unsigned long long* n_clusters_kept;
    unsigned long long* n_hits_kept;
    unsigned long long* dbg; // 8 words, DRPRG_WAVE_DEBUG only: why stage C1 left entries behind
    double fraction;
    uint32_t min_cluster_size;
};

struct ClusterRec {
    uint32_t read, prg_rev, first_pos, last_pos, n, state;
};

struct ClusterArgs {
    const uint64_t* key; // sorted hits
    const uint32_t* val;
    const uint32_t* scan;   // inclusive scan of cluster-head flags
    const uint32_t* cstart; // n_clusters + 1
    const uint32_t* d_n_clusters; // device scalar
    const uint64_t* offsets;
    const uint32_t* prg_min_path_len;
    ClusterRec* clusters;
    uint32_t* order;
    int w;
    double fraction;
    uint32_t min_cluster_size;
    uint32_t* covg;
    uint32_t* prg_reads;
    unsigned long long* n_clusters_kept;
    unsigned long long* n_hits_kept;
};

// optional HIP events recorded on the launch stream immediately around the dominant kernel of a launch sequence
struct KernelTimer {
    hipEvent_t begin = nullptr, end = nullptr;
};

uint32_t sketch_tile_eval(int halo);
uint32_t sketch_n_tiles(uint64_t n_bases, int halo);
hipError_t launch_sketch_probe(const SketchArgs& a, bool wide_hash, hipStream_t stream, KernelTimer timer = {});
// first read that starts at or after the first staged base (tile * t_eval - halo) of every tile
hipError_t launch_tile_first_read(const uint64_t* offsets, uint32_t n_reads, int t_eval, int halo, uint32_t n_tiles, uint32_t* out,
    hipStream_t stream);
// register-resident form of the direct kernel's candidate form (sketch_wave.hip): k = 15, w in {11, 14}; one tile per wave
bool wave_kernel_applies(int k, int w);
uint32_t wave_tile_eval();
uint32_t wave_n_tiles(uint64_t n_bases);
uint32_t wave_n_slices(uint64_t n_bases); // one slice of tile_cap records per workgroup of four tiles
hipError_t launch_sketch_wave(const SketchArgs& a, hipStream_t stream, KernelTimer timer = {});
// tiles of the candidate form of the direct sequence for these parameters (whichever kernel serves them)
uint32_t direct_candidate_tiles(uint64_t n_bases, int halo, int k, int w, bool wide_hash); // = slices
bool direct_uses_wave_form(int k, int w, bool wide_hash); // sketch_wave_kernel serves these parameters
uint32_t direct_first_read_tiles(uint64_t n_bases, int halo, int k, int w, bool wide_hash); // entries of tile_first_read
// filtered form (k <= 15, w <= 16): bloom = 2^bloom_wbits words of index k-mer codes
uint32_t filter_n_tiles(uint64_t n_bases, int positions_per_lane = 32);
uint32_t filter_grid(bool level0, int n_cus, uint32_t n_tiles);
// device copies of FlatIndex::bloom / bloom0 (bloom0 == nullptr: no level 0)
struct BloomTables {
    const uint32_t* bloom;
    uint32_t bloom_wbits;
    const uint32_t* bloom0;
    uint32_t bloom0_wbits;
    const uint32_t* bloomr; // second stage of the level-0 form (2^BLOOMR_WBITS words)
    const uint32_t* bloom0f; // level 0 and the second-stage bits in one array (the form with the second stage inside the streaming kernel)
    // middle tier (FlatIndex::mid0 / mid_bitmap / midc; nullptr: absent)
    const uint32_t* mid0 = nullptr;
    const uint32_t* mid_bitmap = nullptr;
    const uint32_t* midc = nullptr;
    uint32_t midc_wbits = 0, mid0_bits = 0;
    // small tier, second stage in the L2 (round 6; FlatIndex::blkc): split-block filter of the codes, block chosen by the group's plain 12-mer
    const uint32_t* blkc = nullptr;
    uint32_t blkc_wbits = 0;
};
// Scratch of the filtered launch sequence.  raw_pos: raw_capacity candidate positions (one slice per filter wave);
// cand_info / cand_pos1: raw_capacity entries each; small: filter_small_words() u32; max_len: device scalar that
// receives the length of the longest read holding a minimizer hit.  Overflow bit 2 (value 4) in a.overflow: a slice
// was too small.  The hits are written ordered by (read, position); a.n_hits receives their number.
struct FilterBuffers {
    uint64_t* raw_pos;
    uint4* raw_grp; // raw_capacity entries: level-0 survivors (groups of four positions) on their way to refine_kernel; nullptr unless group_records_requested()
    uint64_t* cand_gp; // raw_capacity entries: the dense, ordered list of candidate positions (cand_gather_kernel); nullptr unless gathered_list_requested()
    uint64_t* cand_info;
    uint32_t* cand_pos1;
    uint4* cand_rec; // raw_capacity entries
    uint64_t raw_capacity;
    uint32_t* small;
    unsigned long long* max_len;
    unsigned long long* stat = nullptr; // middle tier, DRPRG_FT_STATS=1: four device counters (FilterWork::stat)
    // sketch_filter_kernel's tile shares (FilterWork::wave_share) for this launch, or nullptr: its built-in ones; and five zeroed device words that
    // receive ~(earliest start) and the latest end of each of the four wave classes on the 100 MHz wall clock (reported by bench.py; with the
    // static schedule the host sets the next batch's shares by them: mapper.cpp tune_filter_shares)
    const uint32_t* wave_share = nullptr;
    unsigned long long* class_clock = nullptr;
};
// How sketch_filter_kernel hands its wave tiles out (round 6).  Every workgroup owns a contiguous range of the window's tiles (an even split).
// Round 0 is static: wave i of the workgroup owns chunk i, the tiles of 16 * tpw0 split by FilterWork::wave_share (in wave order).  Rounds
// 1 .. n_rounds - 1 are dynamic: ticket k >= first_ticket[r] (drawn from a counter in the workgroup's LDS when the wave's prefetch reaches the
// end of its chunk) is the chunk of size[r] tiles that starts at tile first_tile[r] + (k - first_ticket[r]) * size[r] of the workgroup's range; the sizes
// fall from round to round so that the waves -- which do NOT run at one speed: a SIMD issues for its oldest wave first -- end together
// whatever their speeds are.  Chunk = slice: chunk k of workgroup b fills slice b * per_wg + k, and the chunks are numbered in position
// order, so the slices in order are the ordered candidate list.  A workgroup's last chunk runs to the end of its range.  Every chunk of a
// dynamic schedule holds at least two tiles that lie wholly inside the buffer (the kernel's prefetch runs two tiles ahead, into the next
// chunk at most); per_wg == 16: static only, one chunk per wave as until round 5.
constexpr int FT_MAX_ROUNDS = 8;
struct FilterSched {
    uint32_t tpw0;       // round 0: tiles per wave of an even split (0: the kernel divides its window itself -- a read range the host has no tile numbers for)
    uint32_t per_wg;     // chunks = slices per workgroup
    uint32_t n_tiles;    // the tiles the schedule was made for (a dynamic one: the kernel's window must be exactly tiles [0, n_tiles))
    uint32_t first_ticket[FT_MAX_ROUNDS], first_tile[FT_MAX_ROUNDS], size[FT_MAX_ROUNDS]; // [0] unused; first_ticket[r] = 0xFFFFFFFF for r >= n_rounds
    uint32_t n_rounds;
    uint32_t lds_word;   // the workgroup's ticket counter: this word of the kernel's dynamic LDS (set by the launcher; next ticket = 16 + its value)
};
// the schedule of one workgroup for n_tiles wave tiles on n_wg workgroups (window_known: the host knows that the window is tiles [0, n_tiles));
// DRPRG_FT_SCHED=static | f,d,m: share of round 0 in 1/256 of the tiles, divisor of the dynamic rounds (x 16), smallest chunk -- measurements
FilterSched make_filter_sched(uint32_t n_tiles, uint32_t n_wg, bool window_known, const uint32_t share[4]);
// device view of the workspace of one filtered launch sequence (filled by launch_sketch_filter)
struct FilterWork {
    const uint32_t* bloom;
    uint32_t bloom_wbits;
    const uint32_t* bloom0;  // level 0 (nullptr / 0: absent)
    uint32_t bloom0_wbits;
    const uint32_t* bloomr;  // second stage of the level-0 form
    const uint32_t* mid_bitmap; // middle tier: exact bitmap of the canonical index 12-mers (2^24 bits, global memory)
    const uint32_t* midc;       // middle tier: split-block Bloom filter of the index k-mer codes (2^midc_wbits blocks of 16 bytes, global memory)
    uint32_t midc_wbits;
    unsigned long long* stat;   // DRPRG_FT_STATS=1 (middle tier): groups tested, past level 0, past the bitmap, candidate positions
    uint32_t read_begin, read_end; // this launch sequence maps reads [read_begin, read_end) of the batch: the filter kernel
                             // streams the wave tiles (FT_WPOS positions each) that cover their bases, candidates
                             // outside [offsets[read_begin], offsets[read_end]) are dropped by verify_count_kernel
    uint32_t n_slices;       // slices of the candidate buffers (= chunks of the filter kernel's schedule: its workgroups x sched.per_wg)
    // Where a slice lives (round 6: the chunks are not of one size, so neither are their slices): workgroup b's slices share slice_budget
    // entries from b * slice_budget; chunk k of it, tiles [first, end) of the workgroup's range [lo, ..), starts (first - lo) * slice_cpt +
    // k * slice_slack entries in and holds (end - first) * slice_cpt + slice_slack -- room in proportion to its tiles plus a floor for the
    // small ones (a 4-tile chunk that meets five reads from the panel).  The filter kernel leaves start and room next to the count.
    uint32_t slice_budget, slice_cpt, slice_slack;
    uint64_t* raw_pos;       // global base position of a candidate k-mer, ascending per slice
    uint32_t* slice_count;   // [n_slices], clamped to the slice's room (more: overflow bit 2, the host runs the batch again)
    uint32_t* slice_base;    // [n_slices]: first entry of the slice in raw_pos (/ raw_grp)
    uint32_t* slice_cap;     // [n_slices]: its room
    uint32_t* super_count;   // [MAX_SLICES], zero before the launch (counters_home_kernel clears it behind every sequence): the candidates of
                             // slices 8 s .. 8 s + 7 (what verify_scan_kernel scans)
    FilterSched sched;       // sketch_filter_kernel's chunk schedule
    uint4* raw_grp;          // level-0 form only, the slices' geometry: {position of a surviving group of four k-mers (lo, hi),
                             // its 16 bases, the 2 after them}, ascending per slice; refine_kernel turns them into raw_pos
    uint32_t* grp_count;     // [n_slices], clamped like slice_count
    uint32_t* cand_prefix;   // [n_slices + 1]: exclusive scan of the clamped counts
    const uint32_t* cand_total; // the number of candidates (filtered sequence: &cand_prefix[n_slices])
    uint64_t* cand_gp;       // [candidates]: global base position of the candidate k-mer, ascending (the slices gathered; nothing writes
                             // it after cand_gather_kernel: read_verify_kernel's workgroups read their neighbours' entries)
    uint64_t* cand_info;     // [candidates]: slot << 32 | strand << 31 | read
    uint32_t* cand_pos1;     // [candidates]: read position + 1 of a minimizer, 0 = not a minimizer
    uint4* cand_rec;         // [candidates]: what read_cluster_kernel needs of a minimizer: first index record, number of
                             // records, group << 16 | size threshold, coverage index of the first record (0,0,.. = not a minimizer)
    uint32_t ex_grid;        // workgroups of verify_count_kernel / expand_kernel
    uint32_t verify_grid;    // workgroups of the verification kernel of this batch (verify_count_kernel: ex_grid; read_verify_kernel: its
                             // persistent grid): as many words of wg_hits / wg_nmin / wg_maxlen hold its totals
    uint32_t *wg_hits, *wg_nmin, *wg_maxlen, *wg_base; // [ex_grid]
    unsigned long long* max_len; // longest read that holds a minimizer hit (this batch)
    unsigned long long* class_clock; // FilterBuffers::class_clock (may be null)
    uint32_t wave_share[4];  // sketch_filter_kernel: tiles of the waves 4c .. 4c + 3 of a workgroup, in 1/256 of an even share (sum 1024); see its launch
    uint32_t verify_interleave; // verify_scan_kernel: rounds of 64 candidates dealt out over the workgroups instead of one stretch of the list each
    uint32_t debug;          // ablation switches for profiling (DRPRG_FT_DEBUG): 1 = skip the Bloom test, 8 = every read through the
                             // generic pipeline, 16 / 32 = verify_count_kernel without its window scan / table probe and
                             // everything after it (wrong results: timing only, tools/dbg16.sh)
};
constexpr uint32_t RC_WAVE_MAX_WG = 1024; // workgroups of read_cluster_wave_kernel at most
constexpr uint32_t RC_CHUNK_OWN = 1536; // candidates a chunk of read_cluster_kernel owns (read_cluster.hip RC_OWN)
constexpr uint32_t READ_NONE = 0x7FFFFFFFu; // "read" of a candidate that lies past the last whole k-mer of the buffer

// per-read clustering straight from the candidate list (read_cluster_kernel)
struct ReadClusterArgs {
    const uint32_t* prg_min_path_len;
    double fraction;
    uint32_t min_cluster_size;
    int max_diff;
    uint32_t n_prgs;
    uint32_t* covg;
    uint32_t* prg_reads;
    unsigned long long* n_clusters_kept;
    unsigned long long* n_hits_kept;
    unsigned long long* n_complex; // reads left to the generic pipeline (their candidates keep cand_pos1 != 0)
    uint32_t* chunk_counter;       // zeroed device scalar: work distribution of read_cluster_kernel
    // candidates still in their tile slices (direct sequence without tile_gather_kernel): slice_prefix != nullptr.  The ordered
    // list's entry d lives in slice s = the one with slice_prefix[s] <= d < slice_prefix[s + 1], at s * a.tile_cap + (d - slice_prefix[s]);
    // handled candidates are marked cand_pos1[d] = mark_epoch in the (otherwise unused) dense array
    const uint32_t* slice_prefix; // [n_slices + 1], exclusive scan of the slice counts
    uint32_t n_slices, mark_epoch;
    const uint32_t* block_first;  // [total / 64 + 1]: the slice that holds entry 64 m of the ordered list (tile_totals_kernel writes it into the
                                  // -- until a dense list is gathered -- unused fw.cand_info)
    // the wave form (read_cluster_wave.hip) runs first and counts in *n_unfit the reads it leaves untouched (long reads, minimizers with many
    // index records); read_cluster_kernel then runs as a SECOND PASS over what is left: second_pass != 0 makes it return at once when
    // *n_unfit == 0 and skip the candidates that are handled already
    unsigned long long* n_unfit;
    int second_pass;
    uint32_t* wg_partials; // [RC_WAVE_MAX_WG][n_prgs + 4]: per-workgroup histogram and counters of the wave form (summed by its last workgroup)
    uint32_t* wg_done;     // zero before the launch: workgroups of the wave form that have finished
    uint32_t* chunk_flags; // [candidate capacity / RC_CHUNK_OWN + 2], zero before the launch: the wave form sets word c when it leaves a read whose first
                           // candidate lies in read_cluster_kernel's chunk c; the second pass takes only those chunks
    // the batch totals of the candidate stage (hits, minimizers, longest read with a hit) summed by workgroup 0 of read_cluster_kernel
    // from verify_count_kernel's per-workgroup words instead of by a kernel of their own (hit_scan_kernel: 6 us of launch + one round
    // trip; it still runs when the hits are counted again for the generic pipeline, which also needs its prefix sums)
    const uint32_t *wg_hits, *wg_nmin, *wg_maxlen;
    uint32_t n_wg;                 // 0: the totals are somebody else's business
    unsigned long long *tot_hits, *tot_minimizers, *tot_max_len;
    const uint32_t* overflow_word; // bit 2: a candidate slice overflowed (the host runs the batch again: minimizers must not count twice)
    unsigned long long* phase_clock; // DRPRG_RC_DEBUG=1: 12 counters, clock cycles thread 0 of every workgroup spent per phase (else null)
    uint32_t minpath_in_lds;         // set by launch_read_cluster: the dynamic LDS holds [n_prgs] u16 shortest paths behind the histogram
};
// DRPRG_RC_FORM=wave (read at every call): launch_read_cluster runs the wave form first; its flag words must be zero before the launch
bool read_cluster_wave_form_requested();
// from[0 .. n) -> to[0 .. n) (the device address of pinned host memory), then from[0 .. n) = 0; n <= 64.  zero != nullptr: zero[0 .. n_zero)
// = 0 as well (n_zero a multiple of 4, zero 16-byte aligned: the superblock counts of the filtered sequence)
hipError_t launch_counters_home(unsigned long long* from, unsigned long long* to, uint32_t n, hipStream_t stream, uint32_t* zero = nullptr, uint32_t n_zero = 0);
uint32_t* filter_super_counts(uint32_t* small); // the superblock counts inside a FilterBuffers::small block ...
uint32_t filter_super_words();                  // ... and how many words they are
bool group_records_requested();  // ... will want FilterBuffers::raw_grp (DRPRG_FILTER_FORM=refine, experimental library)
bool gathered_list_requested(); // this launch of the filtered sequence will want FilterBuffers::cand_gp (DRPRG_VERIFY_FORM=gather / read)
size_t filter_small_words();
// the fields of fw that the consumers of a dense candidate list use (candidates.hip, read_cluster.hip)
void init_candidate_work(FilterWork& fw, const FilterBuffers& b, int n_cus);
// filter -> candidates -> verify -> per-read clustering of the reads that fit read_cluster_kernel (coverage, PRG read
// counts and the kept-cluster counters are updated); a.n_hits receives the number of hits of the whole batch,
// rc.n_complex the number of reads left over.  fw is filled for the two follow-up calls.
// Reads [read_begin, read_end) of the batch only (their bases are located on the device): the host can run several such
// sequences, each with its own FilterBuffers and scratch counters, on different streams.
hipError_t launch_sketch_filter(const SketchArgs& a, uint32_t read_begin, uint32_t read_end, const BloomTables& bt, int n_cus,
    const FilterBuffers& b, const ReadClusterArgs& rc, FilterWork& fw, hipStream_t stream, KernelTimer timer = {});
// leftover reads: a.n_hits receives the number of their hits, b.max_len their longest read ...
hipError_t launch_filter_recount(const SketchArgs& a, const FilterWork& fw, hipStream_t stream);
// ... and their hits are written to a.hit_key / a.hit_val ordered by (read, position)
hipError_t launch_filter_expand(const SketchArgs& a, const FilterWork& fw, hipStream_t stream);
// Candidate form of the direct sequence: launch_sketch_probe with a.tile_cap != 0, then the tile slices -> one dense
// ordered candidate list (fw.cand_info / cand_pos1 / cand_rec, *fw.cand_total) -> read_cluster_kernel.  tile_prefix:
// n_tiles + 1 words; temp: scan_temp_bytes(n_tiles + 1) bytes.  a.n_hits receives the hits of the batch; overflow bit 2:
// a tile slice or the dense list (dense_capacity entries) was too small (nothing was counted then).
hipError_t launch_direct_candidates(const SketchArgs& a, bool wide_hash, uint32_t* tile_prefix, void* temp, size_t temp_bytes,
    uint64_t dense_capacity, const ReadClusterArgs& rc, int n_cus, FilterWork& fw, hipStream_t stream, KernelTimer timer = {},
    uint32_t slices_mark = 0); // slices_mark != 0: no gathered list, read_cluster_kernel reads the slices and marks handled candidates with it
// the gathered list after such a batch, for the reads that were left over (handled candidates get position 0)
hipError_t launch_tile_gather_marked(const SketchArgs& a, const FilterWork& fw, const uint32_t* tile_prefix, uint32_t n_tiles, uint64_t dense_capacity,
    uint32_t mark, hipStream_t stream);
hipError_t exclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t stream);
// hits ordered by (read, pos) -> ordered by (read, prg, strand, pos), in place; meant for short reads.  scratch: u32
// words (>= n) for the list of reads that need reordering; count: zeroed device scalar
hipError_t launch_read_sort(uint64_t* key, uint32_t* val, uint32_t n, uint32_t* scratch, uint64_t scratch_words, unsigned long long* count,
    hipStream_t stream);
size_t sort_temp_bytes(uint32_t n);
size_t scan_temp_bytes(uint32_t n);
hipError_t sort_hits(void* temp, size_t temp_bytes, const uint64_t* key_in, uint64_t* key_out, const uint32_t* val_in,
    uint32_t* val_out, uint32_t n, hipStream_t stream);
hipError_t launch_cluster_flags(const uint64_t* key, uint32_t n, int max_diff, uint32_t* head, uint32_t* scan, void* temp,
    size_t temp_bytes, hipStream_t stream);
hipError_t launch_cluster_starts(const uint32_t* head, const uint32_t* scan, uint32_t n, uint32_t* cstart, hipStream_t stream);
hipError_t launch_cluster_pipeline(const ClusterArgs& a, uint32_t n_hits, uint32_t n_prgs, hipStream_t stream);
hipError_t launch_vector_add_u32(uint32_t* dst, const uint32_t* src, uint64_t n, hipStream_t stream); // dst[i] += src[i]
// packed.hip: 2-bit packed reads -> ASCII (A C G T, 'N' at the positions in npos and in the 64 bytes behind the last base): what the
// direct sketch kernels and the anchor scan read.  out: n_bases + 64 bytes, 16-byte aligned.
hipError_t launch_unpack(const uint32_t* words, uint64_t n_bases, const uint64_t* npos, uint64_t n_npos, uint8_t* out, hipStream_t stream);
// bits[i >> 4] |= 1 << (i & 15) for every position i of npos below n_bases (bits: zeroed u16[ceil(n_bases / 16)], 4-byte aligned)
hipError_t launch_mark_npos(const uint64_t* npos, uint64_t n_npos, uint64_t n_bases, uint16_t* bits, hipStream_t stream);
// ... and ASCII -> packed on the device (harnesses: bench.py packs its synthetic batch with it).  words: ceil(n_bases / 16); npos: room
// for npos_cap positions, *n_npos (zeroed by the caller) counts all of them (more than npos_cap: overflow); positions come out
// unordered: sort them before use.
hipError_t launch_pack(const uint8_t* bases, uint64_t n_bases, uint32_t* words, uint64_t* npos, uint64_t npos_cap, unsigned long long* n_npos,
    hipStream_t stream);

// anchor_scan.hip: reads of a resident batch that hold one of the (sorted) anchor k-mers of length A -- every such read once,
// in any order, appended to `list` (count keeps counting past list_cap).  prefilter: 2^16 bits, bit (kmer & 0xFFFF) set for every
// anchor; flags: n_reads words, zero before the launch.
struct SelectedRead {
    uint64_t offset;       // of its first base in the batch's base array
    uint32_t len, read, batch, pad;
};
hipError_t launch_anchor_scan(const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, uint64_t n_bases, const uint64_t* anchors,
    uint32_t n_anchors, uint32_t A, const uint32_t* prefilter, uint32_t batch, uint32_t* flags, unsigned long long* count, SelectedRead* list,
    uint64_t list_cap, int n_cus, hipStream_t stream);
struct GatherEntry {
    const uint8_t* src;
    uint64_t dst;
    uint32_t len, pad;
};
hipError_t launch_gather_reads(const GatherEntry* table, uint32_t n, uint8_t* out, hipStream_t stream); // out[dst .. dst+len) = src[0 .. len)

} // namespace dev
} // namespace drprg
