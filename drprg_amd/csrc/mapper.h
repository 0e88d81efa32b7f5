// mapper.h -- device context of the predict hot path: index tables resident in HBM, per-batch
// workspace, and the launch sequence sketch+probe -> sort -> cluster -> accumulate.
//
// This is the MI355X replacement for the read loop of `pandora map` (pangraph_from_read_file),
// spawned by the reference at /root/reference/src/lib.rs:580-642.
#pragma once
#include "index.h"
#include "kernels.h"
#include <vector>

namespace drprg {

struct MapCounters {
    uint64_t reads = 0, bases = 0, minimizers = 0, hits = 0, clusters_kept = 0, hits_kept = 0;
    uint64_t kernel = 0; // sketch kernel in use: 1 direct (sketch_probe_kernel), 2 Bloom-prefiltered (sketch_filter_kernel)
    uint64_t leftover_reads = 0; // filtered sequence: reads that read_cluster_kernel left to the generic pipeline
};

class Mapper {
public:
    Mapper(const FlatIndex& idx, const MapParams& p, int device);
    ~Mapper();
    Mapper(const Mapper&) = delete;
    Mapper& operator=(const Mapper&) = delete;

    int device() const { return device_; }
    uint32_t n_knodes() const { return n_knodes_; }
    uint32_t n_prgs() const { return n_prgs_; }
    hipStream_t stream() const { return stream_; }
    void set_params(const MapParams& p);

    // Map one batch that is already resident in HBM.  d_bases must be 16-byte aligned.  Coverage is
    // accumulated into d_covg (u32[2*n_knodes]) and d_prg_reads (u32[n_prgs]); nullptr selects the
    // context's own accumulators.  Asynchronous on `stream` except for one 16-byte read-back of the
    // hit count between the sketch and the sort.
    void map_device(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
        uint32_t* d_covg, uint32_t* d_prg_reads, hipStream_t stream);

    // The same batch without the host waiting for it: the launch sequence is queued on `stream` and the call returns; the
    // read-back the sequence ends with (buffer overflow flags, reads left to the generic pipeline, counters) is looked at
    // while the NEXT batch runs -- by the next map_device_async, or by sync() and everything that reads results -- so that
    // back-to-back batches leave no gap on the device (measured on 10 M x 150 bp: ~50 us of host round trip per 0.7 ms
    // batch).  The caller keeps d_bases / d_offsets / the accumulators of a batch valid and unchanged until the call after
    // the next one returns, or sync() does (a batch whose candidate buffers overflowed is run again from them).
    // Filtered sequence with one lane, and the direct sequence in its candidate form (two tile workspaces in turn); everything else
    // falls back to map_device.
    void map_device_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
        uint32_t* d_covg, uint32_t* d_prg_reads, hipStream_t stream);
    void sync(); // completes the batch map_device_async left in flight and waits for its stream
    // The same two entries for a batch in the 2-bit packed form (kernels.h SketchArgs::packed): d_words = u32[ceil(n_bases / 16)], 8-byte
    // aligned; d_npos = the n_npos ascending positions of the bases that are not ACGTacgt (may be null when n_npos == 0).  The filtered
    // sequence reads the words directly; the direct sequences get an ASCII expansion made on the device (packed.hip).
    void map_device_packed(const uint32_t* d_words, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, const uint64_t* d_npos, uint64_t n_npos,
        uint32_t* d_covg, uint32_t* d_prg_reads, hipStream_t stream, bool deferred);

    // Map a host batch (copies through pinned staging buffers, then map_device on the own accumulators).
    void map_host(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads);
    // The same without waiting for the kernels: the copy to the device runs on a copy stream into one of two staging sets, the
    // launch sequence is queued behind it (map_device_async) and the call returns as soon as the COPY is done -- the host block
    // may be reused then -- so that the next block's copy overlaps this block's kernels.  sync() (or anything that reads
    // results) completes what is in flight.  Sequences without a deferred form fall back to map_host.
    void map_host_async(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads);
    // ASCII -> packed on the device (harnesses: bench.py packs its synthetic batch; tests).  Returns the number of non-ACGT bases; their
    // positions are in d_npos, ascending, if there are at most npos_cap of them.  Synchronises `stream`.
    uint64_t pack_on_device(const uint8_t* d_bases, uint64_t n_bases, uint32_t* d_words, uint64_t* d_npos, uint64_t npos_cap, hipStream_t stream);
    // a host batch in either form (packed: `bases` points at the words, npos / n_npos as above, in host memory)
    struct HostBatch {
        const uint8_t* bases = nullptr;
        const uint64_t* offsets = nullptr;
        uint64_t n_reads = 0;
        bool packed = false;
        const uint64_t* npos = nullptr;
        uint64_t n_npos = 0;
        uint64_t n_bases() const { return offsets[n_reads]; }
        uint64_t payload_bytes() const { return packed ? ((n_bases() + 15) / 16) * 4 : n_bases(); }
    };
    void map_host(const HostBatch& b);
    void map_host_async(const HostBatch& b);
    // Reads that stay in HBM (off by default).  keep_reads(max_bytes > 0): from now on map_host_async copies every block into
    // device memory of its own instead of a staging set and leaves it there -- at most max_bytes of it; one byte more and
    // everything kept is dropped and the staging sets are back.  kept_complete(): every read mapped since keep_reads() /
    // reset_coverage() is among kept().  What they are for: select_reads_with_anchors below (the pile-up of `discover` without a
    // second pass over the file) and mapping the same reads again against another index (map_kept_from).
    struct KeptBatch {
        const uint8_t* d_bases; // (packed: the words)
        const uint64_t* d_offsets;
        uint64_t n_reads, n_bases;
        bool packed = false;
        const uint64_t* d_npos = nullptr;
        uint64_t n_npos = 0;
    };
    void keep_reads(uint64_t max_bytes);
    bool kept_complete() const { return kept_cap_ > 0 && !kept_broken_; }
    const std::vector<KeptBatch>& kept() const { return kept_; }
    uint64_t kept_bytes() const { return kept_bytes_; }
    void drop_kept();
    // Every kept read that holds one of `anchors` (k-mers of length A <= 31 packed 2 bits per base, A=0 .. T=3, first base in the
    // high bits) -- and possibly a few that do not -- appended to bases / offsets (offsets[0] = 0 is written when offsets is
    // empty), in batch and read order.  anchor_scan.hip.
    void select_reads_with_anchors(std::vector<uint64_t> anchors, uint32_t A, std::vector<uint8_t>& bases, std::vector<uint64_t>& offsets);
    // maps the batches another Mapper of the same device keeps (it must outlive the call); returns the reads mapped
    uint64_t map_kept_from(const Mapper& other);

    // this += other, on the device (the other Mapper's vectors are left as they are): peer copy into a scratch buffer + one
    // add kernel, or the add kernel alone when both live on the same device.  Both are synchronised first.
    void add_vectors_from(Mapper& other);

    // page-locked host memory for ingest blocks (H2D copies from it run at DMA speed)
    static void* pinned_alloc(size_t bytes);
    static void pinned_free(void* p);
    static void warm_device(int device);

    void reset_coverage(bool new_sample = true); // new_sample: the reads kept in HBM (keep_reads) go as well
    // own accumulators
    uint32_t* d_covg() const { return d_covg_; }
    uint32_t* d_prg_reads() const { return d_prg_reads_; }
    void download(std::vector<uint32_t>& covg, std::vector<uint32_t>& prg_reads);
    void upload(const std::vector<uint32_t>& covg, const std::vector<uint32_t>& prg_reads);
    MapCounters counters(); // synchronises
    // out[0] = words of the L2-resident Bloom tier in front of the probe table (0: none), out[1] = bytes of the probe table
    // (keys + slot records), out[2] = bytes of the LDS-resident filter arrays (0: none), out[3] = sequence in use (1/2/3)
    void device_tables(uint64_t out[6]) const;

    // timing of the dominant kernel (HIP events on the launch stream), for bench.py
    // is the batch queued by map_device_async (not yet completed) accumulating into these buffers?
    bool pending_writes_to(const uint32_t* covg, const uint32_t* prg_reads) const
    {
        return pending_.active && ((covg && pending_.covg == covg) || (prg_reads && pending_.prg_reads == prg_reads));
    }
    // sketch_filter_kernel's schedule and how it went in the batch completed last (bench.py; drprg_hip_filter_schedule): out[0] rounds (1: static),
    // [1] slices = tickets, [2] tiles per wave of round 0 (even share), [3..6] shares of the four wave classes in 1/256 of an even one,
    // [7..10] when the classes were through in 10 ns from the kernel's first wave (0: not clocked), [11] slices per workgroup, [12..19] chunk size per round
    void filter_schedule(uint64_t out[20]);
    void enable_kernel_timing(bool on) { timing_ = on; }
    double sketch_ms_total() const { return sketch_ms_; }
    uint64_t sketch_launches() const { return sketch_launches_; }
    void reset_kernel_timing() { sketch_ms_ = 0; sketch_launches_ = 0; }

private:
    void zero_now(void* p, int value, size_t bytes);
    void ensure_workspace(uint64_t hit_capacity);
    // One filtered launch sequence (sketch_filter -> refine -> candidates -> read_cluster) and everything private to it.
    // A batch is cut into as many read ranges as there are lanes (default: one); lane 0 runs on the caller's stream, the
    // others on their own, so that the candidate / cluster kernels of one range can overlap the filter kernel of the next.
    struct Lane {
        hipStream_t stream = nullptr; // lanes >= 1 (lane 0 uses the stream of the call)
        hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr;
        uint64_t raw_capacity = 0;
        uint64_t *raw_pos = nullptr, *cand_gp = nullptr, *cand_info = nullptr;
        uint4 *raw_grp = nullptr, *cand_rec = nullptr;
        uint32_t *cand_pos1 = nullptr, *small = nullptr;
        uint32_t* rc_partials = nullptr; // RC_WAVE_MAX_WG x (n_prgs + 4) words: per-workgroup sums of read_cluster_wave_kernel
        uint32_t* rc_flags = nullptr; // raw_capacity / RC_CHUNK_OWN + 2 words: chunks in which read_cluster_wave_kernel left a read (zeroed per batch)
        unsigned long long* d_scratch = nullptr; // L_N x u64 per-sequence counters (below)
        unsigned long long* h_scratch = nullptr; // pinned mirror
        unsigned long long* h_scratch_dev = nullptr; // its device address
        bool scratch_zero = false;               // d_scratch is known to be zero on the device
        bool ft_packed = false;                  // the batch of the sequence in flight is a packed one (tune_filter_shares)
        dev::FilterWork fw {};
        uint32_t r0 = 0, r1 = 0;                 // read range of the current batch
    };
    enum { L_HITS = 0, L_OVERFLOW = 1, L_MAXLEN = 2, L_UNSORTED = 3, L_COMPLEX = 4, L_CHUNK = 5, L_MINIMIZERS = 6, L_UNFIT = 7, L_FT_CLOCK = 8 /* five words: FilterBuffers::class_clock */, L_N = 13 };
    // What makes a device batch a packed one (2-bit words instead of bytes): where its non-ACGT positions are.  Passed along with the batch
    // pointer to everything that touches the batch (nullptr: ASCII) and kept in Pending for the deferred completion -- until round 5 the
    // batch's ADDRESS was looked up in a map, which every free and every recycled address had to keep honest.
    struct PackedInfo {
        const uint64_t* d_npos = nullptr;
        uint64_t n_npos = 0;
    };
    void ensure_lanes(int n, uint64_t raw_capacity);
    void grow_lane(Lane& lane, uint64_t raw_capacity);
    void free_lane(Lane& lane);
    void launch_lane(Lane& lane, hipStream_t stream, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads,
        uint64_t n_bases, uint32_t* covg, uint32_t* prg_reads, const PackedInfo* pk);
    void wait_stream(hipStream_t stream);
    void leftovers(Lane& lane, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg,
        uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk);
    void run_batch_direct_candidates(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases,
        uint32_t* covg, uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk);
    void run_batch(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases,
        uint32_t* d_covg, uint32_t* d_prg_reads, hipStream_t stream, const PackedInfo* pk);
    void cluster_hits(const uint64_t* d_offsets, uint32_t n_hits, bool ordered, unsigned long long* d_unsorted, uint32_t* d_covg,
        uint32_t* d_prg_reads, hipStream_t stream);
    dev::SketchArgs sketch_args(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, const PackedInfo* pk) const;
    // d_bases itself, or -- for a packed batch -- its ASCII expansion in scratch buffer `slot` (0 / 1: the two tile sets of the direct
    // sequence's candidate form, 2: the generic sequence), made on `stream`
    const uint8_t* ascii_view(int slot, const uint8_t* d_bases, uint64_t n_bases, hipStream_t stream, const PackedInfo* pk);
    uint8_t* d_unpacked_[3] = { nullptr, nullptr, nullptr };
    unsigned long long* d_pack_count_ = nullptr; // pack_on_device's counter word
    uint64_t unpacked_cap_[3] = { 0, 0, 0 };
    void map_device_impl(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t* d_covg, uint32_t* d_prg_reads,
        hipStream_t stream, const PackedInfo* pk);
    void map_device_async_impl(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t* d_covg, uint32_t* d_prg_reads,
        hipStream_t stream, const PackedInfo* pk);
    void read_counters(hipStream_t stream);
    void note_kernel_time();
    // deferred completion: two lanes take the batches in turn; `pending_` is the batch whose read-back nobody has looked at yet
    struct Pending {
        bool active = false;
        int lane = 0;
        const uint8_t* d_bases = nullptr;
        const uint64_t* d_offsets = nullptr;
        uint32_t n_reads = 0;
        uint64_t n_bases = 0;
        uint32_t *covg = nullptr, *prg_reads = nullptr;
        hipStream_t stream = nullptr;
        bool direct = false; // a batch of the direct sequence's candidate form (lane = its tile set)
        bool packed = false; // the batch is 2-bit words ...
        PackedInfo pk {};    // ... with these non-ACGT positions
    };
    void complete_pending();
    void complete_batch(const Pending& p);
    void finish_lane(Lane& lane, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg,
        uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk);
    Pending pending_;
    std::vector<Lane> pipe_lanes_;
    int pipe_next_ = 0;

    int device_ = 0;
    MapParams params_;
    bool wide_hash_ = false;
    int halo_ = 16;
    uint32_t n_knodes_ = 0, n_prgs_ = 0, table_bits_ = 0;
    hipStream_t stream_ = nullptr;
    // index tables
    void* d_slot_key_ = nullptr;
    uint2* d_slot_rec_ = nullptr;
    uint4* d_slot_first_ = nullptr;
    uint32_t* d_rec_knode_ = nullptr;
    uint16_t* d_rec_prg_ = nullptr;
    uint32_t* d_min_path_len_ = nullptr;
    uint32_t* d_prg_thr_ = nullptr;         // per PRG: floor(shortest k-mer path * cluster fraction), follows set_params
    std::vector<uint32_t> h_min_path_len_;
    uint32_t* d_bloom_ = nullptr;
    uint32_t bloom_wbits_ = 0;
    uint32_t* d_pbloom_ = nullptr; // Bloom tier of the direct kernel (large indexes)
    uint32_t pbloom_wbits_ = 0;
    uint32_t* d_bloom0_ = nullptr; // level 0 of the filter (k = 15, small indexes)
    uint32_t* d_bloom0f_ = nullptr; // level 0 + second-stage bits in one array
    uint32_t bloom0_wbits_ = 0;
    uint32_t* d_bloomr_ = nullptr; // second stage of the level-0 form
    // middle tier of the filter (FlatIndex::mid0 / mid_bitmap / midc): level 0 in LDS, the other two in global memory (L2-resident)
    uint32_t *d_mid0_ = nullptr, *d_mid_bitmap_ = nullptr, *d_midc_ = nullptr;
    uint32_t midc_wbits_ = 0, mid0_bits_ = 0;
    uint32_t* d_blkc_ = nullptr; // small tier: the second stage as a split-block filter in global memory (FlatIndex::blkc)
    uint32_t blkc_wbits_ = 0;
    bool use_mid_ = false;                   // the filtered sequence runs in its middle-tier form
    unsigned long long* d_ft_stat_ = nullptr; // DRPRG_FT_STATS=1: groups tested / past level 0 / past the bitmap, candidate positions
    int n_cus_ = 256;
    bool use_filter_ = false;
    bool fuse_in_kernel_ = false; // sketch_wave_kernel clusters the reads inside one tile itself (DRPRG_WAVE_FUSE=1)
    int fuse_mode_ = 1;          // 2: only reads whose minimizers all have one index record (DRPRG_WAVE_FUSE=2)
    bool use_direct_cands_ = false; // direct sketch kernel in its candidate form (read_cluster_kernel instead of sort + cluster kernels)
    // accumulators
    uint32_t* d_covg_ = nullptr;
    uint32_t* d_prg_reads_ = nullptr;
    unsigned long long* d_counters_ = nullptr; // 8 x u64: hits(batch), minimizers, clusters_kept, hits_kept, overflow, ...
    unsigned long long* h_counters_ = nullptr; // pinned mirror
    // per-batch device counters are read back once per batch and summed here (an aborted attempt -- a buffer that has to
    // grow -- is simply not added, whatever sequence ran before it)
    uint64_t tot_reads_ = 0, tot_bases_ = 0, tot_hits_ = 0, tot_leftover_ = 0, tot_minimizers_ = 0;
    // workspace
    uint64_t hit_capacity_ = 0;
    uint64_t *d_key_a_ = nullptr, *d_key_b_ = nullptr;
    uint32_t *d_val_a_ = nullptr, *d_val_b_ = nullptr;
    uint32_t *d_head_ = nullptr, *d_scan_ = nullptr, *d_cstart_ = nullptr, *d_order_ = nullptr;
    dev::ClusterRec* d_clusters_ = nullptr;
    // candidate workspace of the filtered sequence
    std::vector<Lane> lanes_;
    int max_lanes_ = 1;             // DRPRG_HIP_LANES (1..4).  Measured on configs[1] (10 M x 150 bp): 0.66 ms with one lane, 0.73 / 0.83 ms
                                    // with two / four: a resident sketch_filter_kernel workgroup holds 128 KB of a CU's LDS, so refine
                                    // and read_cluster of the other range wait for it, and what does overlap (verify) competes for the
                                    // same VALU issue slots -- every kernel stretches by about what the overlap saves
    uint64_t lanes_min_bases_ = 64ull << 20; // smaller batches always take one lane
    hipEvent_t ev_begin_ = nullptr; // recorded on the caller's stream: the other lanes start behind it
    void* d_temp_ = nullptr;
    // candidate form of the direct sequence: one slice of `cap` records per tile.  Two sets of the workspace: the synchronous calls use
    // the first; map_device_async takes them in turn (a batch's slices must survive until its read-back has been looked at: reads left
    // to the generic pipeline are gathered from them)
    struct TileSet {
        uint32_t slice_cap = 256, ws_tiles = 0, ws_cap = 0, first_cap = 0;
        uint64_t* d_tile_info = nullptr;
        uint32_t *d_tile_pos1 = nullptr, *d_tile_count = nullptr, *d_tile_hits = nullptr, *d_tile_nmin = nullptr, *d_tile_prefix = nullptr, *d_tile_fast = nullptr;
        uint4* d_tile_rec = nullptr;
        void* d_tile_temp = nullptr;
        size_t tile_temp_bytes = 0;
        uint32_t* d_tile_first = nullptr; // first read of every tile
        // packed batches through sketch_wave_kernel: one bit per base, set for the positions in the batch's npos (allocated with the first
        // batch that has any; all zero except between a launch and the next launch on this set)
        uint16_t* d_nbits = nullptr;
        uint64_t nbits_cap = 0; // u16 words
        bool nbits_dirty = false;
        hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr;
        uint32_t mark = 0, n_tiles = 0;   // of the batch in flight
        dev::SketchArgs a_done {};
    };
    TileSet tsets_[2];
    TileSet* ts_ = &tsets_[0];
    uint32_t slices_epoch_ = 0x80000000u; // mark of the last batch whose candidates read_cluster_kernel took from the slices
    void free_tile_set(TileSet& t);
    void ensure_tile_workspace(TileSet& t, uint32_t n_tiles, uint32_t tile_cap);
    void direct_launch(int set, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg, uint32_t* prg_reads,
        hipStream_t stream, bool timed_by_set_events, const PackedInfo* pk);
    bool direct_finish(int set, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg, uint32_t* prg_reads,
        hipStream_t stream, int attempt, const PackedInfo* pk);
    size_t temp_bytes_ = 0;
    // host staging
    uint8_t* h_bases_ = nullptr;
    uint64_t* h_offsets_ = nullptr;
    uint8_t* d_bases_ = nullptr;
    uint64_t* d_offsets_ = nullptr;
    uint64_t* d_npos_ = nullptr;
    uint64_t stage_bases_cap_ = 0, stage_reads_cap_ = 0, stage_npos_cap_ = 0;
    // map_host_async: two staging sets taken in turn, their copies on a stream of their own
    struct Stage {
        uint8_t* d_bases = nullptr;
        uint64_t* d_offsets = nullptr;
        uint64_t* d_npos = nullptr;
        uint64_t bases_cap = 0, reads_cap = 0, npos_cap = 0;
        hipEvent_t copied = nullptr;
    };
    Stage stage_[2];
    int stage_next_ = 0;
    hipStream_t copy_stream_ = nullptr;
    // sketch_filter_kernel's tile shares, followed from batch to batch ([0] ASCII, [1] packed input; {0}: the launcher's built-in ones so far)
    uint32_t ft_share_[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    bool ft_adapt_ = true;
    uint64_t ft_last_[20] = {};
    void tune_filter_shares(const Lane& lane, bool packed, uint64_t n_bases);
    // keep_reads: device memory in large pieces, handed out front to back
    std::vector<std::pair<void*, size_t>> kept_arenas_;
    uint8_t* arena_at_ = nullptr;
    size_t arena_left_ = 0;
    std::vector<KeptBatch> kept_;
    uint64_t kept_cap_ = 0, kept_bytes_ = 0;
    bool kept_broken_ = false, in_keep_call_ = false;
    hipEvent_t kept_copied_ = nullptr;
    void* arena_take(size_t bytes);
    uint32_t* d_peer_tmp_ = nullptr; // add_vectors_from: the other device's vectors on this device
    // timing
    bool timing_ = false;
    double sketch_ms_ = 0;
    uint64_t sketch_launches_ = 0;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
};

} // namespace drprg
