// mapper.cpp -- device context and per-batch launch sequence.  See mapper.h.
#include "mapper.h"
#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <hip/hip_runtime.h>
#include <sys/mman.h>

namespace drprg {

#define HIPCHK(x)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess)                                                                                  \
            throw Error(e_ == hipErrorOutOfMemory ? DRPRG_ENOMEM : DRPRG_EIO,                                  \
                std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

template <typename T> static void dmalloc(T*& p, size_t n)
{
    p = nullptr;
    if (n == 0) n = 1;
    HIPCHK(hipMalloc((void**)&p, n * sizeof(T)));
}
template <typename T> static void dfree(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// counter slots
enum { C_HITS = 0, C_MINIMIZERS = 1, C_CLUSTERS_KEPT = 2, C_HITS_KEPT = 3, C_OVERFLOW = 4, C_MAXLEN = 5, C_UNSORTED = 6, C_COMPLEX = 7, C_CHUNK = 8, C_N = 16 };
// reads up to this length get their hits reordered per read (read_sort_kernel); longer ones take the radix sort
constexpr uint64_t READ_SORT_MAX_LEN = 512;

Mapper::Mapper(const FlatIndex& idx, const MapParams& p, int device) : device_(device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(DRPRG_ENODEV, "no HIP device is visible: the drprg hot path has no CPU fallback");
    if (device < 0 || device >= ndev) throw Error(DRPRG_ENODEV, "HIP device " + std::to_string(device) + " does not exist");
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    n_prgs_ = (uint32_t)idx.min_path_len.size();
    n_knodes_ = idx.total_knodes();
    table_bits_ = idx.table_bits;
    bloom_wbits_ = idx.bloom_wbits;
    if (n_prgs_ > dev::MAX_PRGS) throw Error(DRPRG_EOVERFLOW, "more than " + std::to_string(dev::MAX_PRGS) + " PRGs");
    set_params(p);

    const size_t nslot = idx.slot_key.size();
    if (wide_hash_) {
        std::vector<uint64_t> k64(nslot);
        for (size_t i = 0; i < nslot; ++i) k64[i] = idx.slot_cnt[i] ? idx.slot_key[i] : ~0ULL; // empty-slot sentinel
        uint64_t* k = nullptr;
        dmalloc(k, nslot);
        HIPCHK(hipMemcpy(k, k64.data(), nslot * sizeof(uint64_t), hipMemcpyHostToDevice));
        d_slot_key_ = k;
    } else {
        std::vector<uint32_t> k32(nslot);
        for (size_t i = 0; i < nslot; ++i) k32[i] = idx.slot_cnt[i] ? (uint32_t)idx.slot_key[i] : 0xFFFFFFFFu; // empty-slot sentinel
        uint32_t* k = nullptr;
        dmalloc(k, nslot);
        HIPCHK(hipMemcpy(k, k32.data(), nslot * sizeof(uint32_t), hipMemcpyHostToDevice));
        d_slot_key_ = k;
    }
    std::vector<uint2> rec(nslot);
    for (size_t i = 0; i < nslot; ++i) rec[i] = make_uint2(idx.slot_off[i], idx.slot_cnt[i]);
    dmalloc(d_slot_rec_, nslot);
    HIPCHK(hipMemcpy(d_slot_rec_, rec.data(), nslot * sizeof(uint2), hipMemcpyHostToDevice));
    {
        std::vector<uint4> first(nslot);
        for (size_t i = 0; i < nslot; ++i) {
            first[i] = make_uint4(idx.slot_off[i], idx.slot_cnt[i], 0u, 0u);
            if (idx.slot_cnt[i]) {
                const uint32_t r = idx.slot_off[i], prg = idx.rec_prg[r];
                if (idx.min_path_len[prg] >= (1u << 20)) throw Error(DRPRG_EOVERFLOW, "a PRG's shortest k-mer path has 2^20 nodes or more");
                first[i].z = (idx.rec_knode_global[r] << 1) | idx.rec_strand[r];
                first[i].w = prg | (idx.min_path_len[prg] << 12);
            }
        }
        dmalloc(d_slot_first_, nslot);
        HIPCHK(hipMemcpy(d_slot_first_, first.data(), nslot * sizeof(uint4), hipMemcpyHostToDevice));
    }
    const size_t nrec = idx.rec_prg.size();
    std::vector<uint32_t> rk(nrec);
    std::vector<uint16_t> rp(nrec);
    for (size_t i = 0; i < nrec; ++i) {
        rk[i] = (idx.rec_knode_global[i] << 1) | idx.rec_strand[i];
        rp[i] = (uint16_t)idx.rec_prg[i];
    }
    if (n_knodes_ >= (1u << 31)) throw Error(DRPRG_EOVERFLOW, "too many k-mer nodes");
    dmalloc(d_rec_knode_, nrec);
    dmalloc(d_rec_prg_, nrec);
    HIPCHK(hipMemcpy(d_rec_knode_, rk.data(), nrec * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_rec_prg_, rp.data(), nrec * sizeof(uint16_t), hipMemcpyHostToDevice));
    dmalloc(d_min_path_len_, (size_t)n_prgs_);
    HIPCHK(hipMemcpy(d_min_path_len_, idx.min_path_len.data(), n_prgs_ * sizeof(uint32_t), hipMemcpyHostToDevice));
    h_min_path_len_ = idx.min_path_len;
    dmalloc(d_prg_thr_, (size_t)n_prgs_);
    // Bloom tier of the direct kernel: only when keys + slot records outgrow an XCD's 4 MB L2 (<= ~4 keys per 32-bit word)
    if (nslot * (size_t)(wide_hash_ ? 16 : 12) > ((size_t)2 << 20)) {
        pbloom_wbits_ = 10;
        while (((size_t)4 << pbloom_wbits_) < idx.keys.size()) ++pbloom_wbits_;
        std::vector<uint32_t> pb((size_t)1 << pbloom_wbits_, 0);
        for (uint64_t key : idx.keys) {
            const uint32_t m = dev::pbloom_mix(key);
            pb[dev::pbloom_word(m, pbloom_wbits_)] |= dev::pbloom_bits(m);
        }
        dmalloc(d_pbloom_, pb.size());
        HIPCHK(hipMemcpy(d_pbloom_, pb.data(), pb.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    bloom_wbits_ = idx.bloom_wbits;
    if (bloom_wbits_) {
        dmalloc(d_bloom_, idx.bloom.size());
        HIPCHK(hipMemcpy(d_bloom_, idx.bloom.data(), idx.bloom.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    bloom0_wbits_ = bloom_wbits_ ? idx.bloom0_wbits : 0;
    if (bloom0_wbits_) {
        dmalloc(d_bloom0_, idx.bloom0.size());
        HIPCHK(hipMemcpy(d_bloom0_, idx.bloom0.data(), idx.bloom0.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        dmalloc(d_bloom0f_, idx.bloom0f.size());
        HIPCHK(hipMemcpy(d_bloom0f_, idx.bloom0f.data(), idx.bloom0f.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        dmalloc(d_bloomr_, idx.bloomr.size());
        HIPCHK(hipMemcpy(d_bloomr_, idx.bloomr.data(), idx.bloomr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (idx.blkc_wbits) { // the second stage for the L2 (round 6)
            blkc_wbits_ = idx.blkc_wbits;
            dmalloc(d_blkc_, idx.blkc.size());
            HIPCHK(hipMemcpy(d_blkc_, idx.blkc.data(), idx.blkc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    if (idx.midc_wbits) {
        const size_t max_records = mid_tier_max_records(); // (common.h: where the direct sequence takes over)
        if (idx.rec_prg.size() <= max_records) {
            midc_wbits_ = idx.midc_wbits;
            mid0_bits_ = idx.mid0_bits;
            dmalloc(d_mid0_, idx.mid0.size());
            HIPCHK(hipMemcpy(d_mid0_, idx.mid0.data(), idx.mid0.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            dmalloc(d_mid_bitmap_, idx.mid_bitmap.size());
            HIPCHK(hipMemcpy(d_mid_bitmap_, idx.mid_bitmap.data(), idx.mid_bitmap.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            dmalloc(d_midc_, idx.midc.size());
            HIPCHK(hipMemcpy(d_midc_, idx.midc.data(), idx.midc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            if (const char* e = std::getenv("DRPRG_FT_STATS"); e && std::atoi(e)) {
                dmalloc(d_ft_stat_, (size_t)4);
                zero_now(d_ft_stat_, 0, 4 * sizeof(unsigned long long));
            }
        }
    }
    HIPCHK(hipDeviceGetAttribute(&n_cus_, hipDeviceAttributeMultiprocessorCount, device_));
    HIPCHK(hipStreamSynchronize(nullptr)); // every table is on the device before a kernel of a non-blocking stream can ask for it
    set_params(p); // again: the kernel choice depends on the filter being available
#ifdef DRPRG_EXPERIMENTAL // (make EXPERIMENTAL=1; the default build has no such instantiation of sketch_wave_kernel)
    {
        const char* f = std::getenv("DRPRG_WAVE_FUSE");
        const char* d = std::getenv("DRPRG_FT_DEBUG");
        // opt-in (DRPRG_WAVE_FUSE=1; 2 = single-record minimizers only): measured on the 500-locus workload the in-kernel
        // clustering takes 84 % of the candidates away from gather + read_cluster_kernel (2.7 -> 2.1 ms) but costs
        // sketch_wave_kernel 1.65 ms (3.84 -> 5.50 ms): 7.6 ms per 10 M reads against 6.9 ms without it
        fuse_in_kernel_ = f && std::atoi(f) != 0 && !(d && (std::atoi(d) & 8));
        fuse_mode_ = f && std::atoi(f) == 2 ? 2 : 1;
    }
#endif
    if (const char* e = std::getenv("DRPRG_HIP_LANES")) max_lanes_ = std::min(4, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("DRPRG_HIP_LANES_MIN_BASES")) lanes_min_bases_ = std::strtoull(e, nullptr, 10); // (tests: 0)
    // ONE allocation [coverage | reads per PRG]: the sample's whole additive state is one contiguous u32 vector, so the
    // collective of the path is a single ncclAllReduce / ncclReduce over it (capi.cpp)
    dmalloc(d_covg_, 2 * (size_t)n_knodes_ + (size_t)n_prgs_);
    d_prg_reads_ = d_covg_ + 2 * (size_t)n_knodes_;
    dmalloc(d_counters_, (size_t)C_N);
    HIPCHK(hipHostMalloc((void**)&h_counters_, C_N * sizeof(unsigned long long), hipHostMallocDefault));
    HIPCHK(hipEventCreate(&ev0_));
    HIPCHK(hipEventCreate(&ev1_));
    reset_coverage();
}

Mapper::~Mapper()
{
    (void)hipSetDevice(device_);
    try {
        sync();
    } catch (...) { // (nothing to report to from a destructor)
    }
    (void)hipDeviceSynchronize();
    for (Lane& lane : pipe_lanes_) free_lane(lane);
    if (stream_) (void)hipStreamSynchronize(stream_);
    dfree(d_slot_rec_); dfree(d_slot_first_); dfree(d_rec_knode_); dfree(d_rec_prg_); dfree(d_min_path_len_); dfree(d_prg_thr_);
    if (d_slot_key_) (void)hipFree(d_slot_key_);
    dfree(d_covg_); d_prg_reads_ = nullptr; dfree(d_counters_);
    for (int i = 0; i < 3; ++i) dfree(d_unpacked_[i]);
    dfree(d_pack_count_);
    dfree(d_npos_);
    dfree(d_key_a_); dfree(d_key_b_); dfree(d_val_a_); dfree(d_val_b_);
    dfree(d_head_); dfree(d_scan_); dfree(d_cstart_); dfree(d_order_); dfree(d_clusters_);
    if (d_temp_) (void)hipFree(d_temp_);
    for (TileSet& t : tsets_) free_tile_set(t);
    dfree(d_bases_); dfree(d_offsets_); dfree(d_bloom_); dfree(d_bloom0_); dfree(d_bloom0f_); dfree(d_bloomr_); dfree(d_pbloom_); dfree(d_mid0_); dfree(d_mid_bitmap_); dfree(d_midc_); dfree(d_blkc_); dfree(d_ft_stat_);
    for (Lane& lane : lanes_) free_lane(lane);
    for (Stage& st : stage_) {
        dfree(st.d_bases); dfree(st.d_offsets); dfree(st.d_npos);
        if (st.copied) (void)hipEventDestroy(st.copied);
    }
    dfree(d_peer_tmp_);
    for (auto& a : kept_arenas_) (void)hipFree(a.first);
    if (kept_copied_) (void)hipEventDestroy(kept_copied_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    if (ev_begin_) (void)hipEventDestroy(ev_begin_);
    if (h_counters_) (void)hipHostFree(h_counters_);
    if (h_bases_) (void)hipHostFree(h_bases_);
    if (h_offsets_) (void)hipHostFree(h_offsets_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

// Zeroes device memory and returns when it IS zero.  hipMemset on device memory runs on the null stream and may return before it
// has run; this class's streams are non-blocking, so nothing they do waits for the null stream -- a freshly allocated cand_pos1
// was zeroed AFTER verify_count_kernel had written the first batch into it (cold start, several contexts busy on one device:
// one run in ten lost 5 % of its clusters; tools/stress_multi.py).
void Mapper::zero_now(void* p, int value, size_t bytes)
{
    (void)value;
    HIPCHK(hipMemsetAsync(p, 0, bytes, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
}

static std::mutex g_pinned_mu;
static std::set<void*> g_pinned_by_runtime; // blocks that came from hipHostMalloc (pinned_alloc's second choice)

// Page-locked memory for the ingest's parser threads.  Portable: they call this with whatever device is current on their
// thread, and a portable registration is page-locked for every device, so the copy engine of the mapper's device takes it at
// DMA speed.  Ordinary memory (2 MB aligned, transparent huge pages asked for), touched by the calling thread and THEN
// registered: measured on the MI355X box for 32 threads x 32 MB, hipHostMalloc pins at 6.7 GB/s (161 ms), touch +
// hipHostRegister takes 5 + 13 ms, and host -> device copies run at the same 56 GB/s from both (tools/mb_pin.cpp,
// profiles/r03/mb_pin.txt).
void* Mapper::pinned_alloc(size_t bytes)
{
    constexpr size_t HUGE = 2u << 20;
    const size_t n = (bytes + HUGE - 1) / HUGE * HUGE;
    void* p = std::aligned_alloc(HUGE, n);
    if (!p) return nullptr;
    (void)madvise(p, n, MADV_HUGEPAGE);
    for (size_t i = 0; i < n; i += 4096) static_cast<volatile char*>(p)[i] = 0;
    if (hipHostRegister(p, n, hipHostRegisterPortable) != hipSuccess) {
        // (a host that refuses the registration -- a limit on locked pages, an old driver --: the runtime's own allocation, slower to get)
        (void)hipGetLastError();
        std::free(p);
        p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        std::lock_guard<std::mutex> g(g_pinned_mu);
        g_pinned_by_runtime.insert(p);
    }
    return p;
}

void Mapper::pinned_free(void* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned_mu);
        if (g_pinned_by_runtime.erase(p)) {
            (void)hipHostFree(p);
            return;
        }
    }
    (void)hipHostUnregister(p);
    std::free(p);
}

// First HIP call of a process: tens to hundreds of milliseconds of runtime start-up.  drprg_hip_open runs this on a thread of
// its own while the index files are read; errors are left to the Mapper constructor, which reports them.
void Mapper::warm_device(int device)
{
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr);
}

void Mapper::set_params(const MapParams& p)
{
    sync(); // (a batch in flight was launched with the previous parameters)
    // validate everything before any state changes: a refused call leaves the previous parameters in force
    if (p.k < 1 || p.k > 31) throw Error(DRPRG_EINVAL, "k must be in [1,31]");
    if (p.w < 1 || p.w > 1024) throw Error(DRPRG_EINVAL, "w must be in [1,1024]");
    const bool filter_ok = (bloom_wbits_ != 0 || midc_wbits_ != 0) && p.k <= 15 && p.w <= 16;
    if (p.kernel_mode < 0 || p.kernel_mode > 3) throw Error(DRPRG_EINVAL, "kernel must be 0 (auto), 1, 2 or 3");
    if (p.kernel_mode == 2 && !filter_ok)
        throw Error(DRPRG_EINVAL, "the Bloom-prefiltered kernel needs k <= 15, w <= 16 and an index small enough for its filter tiers");
    params_ = p;
    wide_hash_ = p.k > 15;
    halo_ = std::max(16, ((p.w - 1 + 15) / 16) * 16);
    if (d_prg_thr_) { // the PRG's share of the cluster size threshold, in the oracle's own arithmetic
        std::vector<uint32_t> thr(n_prgs_);
        const double fraction = params_.cluster_fraction();
        for (uint32_t i = 0; i < n_prgs_; ++i) thr[i] = (uint32_t)((double)h_min_path_len_[i] * fraction);
        HIPCHK(hipSetDevice(device_));
        HIPCHK(hipMemcpy(d_prg_thr_, thr.data(), n_prgs_ * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipStreamSynchronize(nullptr)); // (the copies of this file run on the null stream, which the non-blocking streams do not wait for)
    }
    use_filter_ = p.kernel_mode == 2 || (p.kernel_mode == 0 && filter_ok);
    use_mid_ = use_filter_ && bloom_wbits_ == 0;
    use_direct_cands_ = p.kernel_mode == 3 || (p.kernel_mode == 0 && !filter_ok);
}

void Mapper::reset_coverage(bool new_sample)
{
    sync();
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipMemsetAsync(d_covg_, 0, (2 * (size_t)n_knodes_ + (size_t)n_prgs_) * sizeof(uint32_t), stream_)); // [coverage | reads per PRG]
    HIPCHK(hipMemsetAsync(d_counters_, 0, C_N * sizeof(unsigned long long), stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    tot_reads_ = tot_bases_ = tot_hits_ = tot_leftover_ = tot_minimizers_ = 0;
    for (auto& sh : ft_share_) // (a reset context starts from the launcher's built-in tile shares again: ADVICE r05)
        for (uint32_t& v : sh) v = 0;
    if (new_sample) drop_kept();
}

void Mapper::ensure_workspace(uint64_t cap)
{
    if (cap <= hit_capacity_) return;
    if (cap >= (1ull << 31)) throw Error(DRPRG_EOVERFLOW, "more than 2^31 minimizer hits in one batch; map smaller batches");
    dfree(d_key_a_); dfree(d_key_b_); dfree(d_val_a_); dfree(d_val_b_);
    dfree(d_head_); dfree(d_scan_); dfree(d_cstart_); dfree(d_order_); dfree(d_clusters_);
    if (d_temp_) (void)hipFree(d_temp_);
    d_temp_ = nullptr;
    hit_capacity_ = cap;
    dmalloc(d_key_a_, cap); dmalloc(d_key_b_, cap); dmalloc(d_val_a_, cap); dmalloc(d_val_b_, cap);
    dmalloc(d_head_, cap); dmalloc(d_scan_, cap); dmalloc(d_cstart_, cap + 1); dmalloc(d_order_, cap);
    dmalloc(d_clusters_, cap);
    temp_bytes_ = std::max(dev::sort_temp_bytes((uint32_t)cap), dev::scan_temp_bytes((uint32_t)cap));
    HIPCHK(hipMalloc(&d_temp_, temp_bytes_ ? temp_bytes_ : 1));
}

void Mapper::free_lane(Lane& lane)
{
    dfree(lane.raw_pos); dfree(lane.raw_grp); dfree(lane.cand_gp); dfree(lane.cand_info); dfree(lane.cand_pos1); dfree(lane.cand_rec); dfree(lane.small); dfree(lane.rc_flags); dfree(lane.rc_partials);
    dfree(lane.d_scratch);
    if (lane.h_scratch) (void)hipHostFree(lane.h_scratch);
    lane.h_scratch = nullptr;
    lane.h_scratch_dev = nullptr;
    if (lane.stream) (void)hipStreamDestroy(lane.stream);
    for (hipEvent_t* e : { &lane.done, &lane.t0, &lane.t1 })
        if (*e) (void)hipEventDestroy(*e);
    lane = Lane();
}

void Mapper::grow_lane(Lane& lane, uint64_t cap)
{
    if (cap <= lane.raw_capacity) return;
    if (cap >= (1ull << 31)) throw Error(DRPRG_EOVERFLOW, "more than 2^31 candidate k-mers in one read range; map smaller batches");
    dfree(lane.raw_pos); dfree(lane.raw_grp); dfree(lane.cand_gp); dfree(lane.cand_info); dfree(lane.cand_pos1); dfree(lane.cand_rec); dfree(lane.rc_flags);
    lane.raw_capacity = cap;
    dmalloc(lane.raw_pos, cap); dmalloc(lane.cand_info, cap); dmalloc(lane.cand_pos1, cap); dmalloc(lane.cand_rec, cap); // (cand_gp: launch_lane, on demand)
    dmalloc(lane.rc_flags, (size_t)(cap / dev::RC_CHUNK_OWN + 3));
    if (!lane.rc_partials) dmalloc(lane.rc_partials, (size_t)dev::RC_WAVE_MAX_WG * ((size_t)n_prgs_ + 4));
    // the slices form of the direct sequence uses cand_pos1 as an array of "handled" marks (mark = a batch's epoch): fresh device
    // memory may hold anything, including a value some later epoch of this or an earlier Mapper takes
    zero_now(lane.cand_pos1, 0, cap * sizeof(uint32_t));
}

void Mapper::ensure_lanes(int n, uint64_t cap)
{
    if (!ev_begin_) HIPCHK(hipEventCreateWithFlags(&ev_begin_, hipEventDisableTiming));
    while ((int)lanes_.size() < n) {
        lanes_.emplace_back();
        Lane& lane = lanes_.back();
        if (lanes_.size() > 1) HIPCHK(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&lane.done, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&lane.t0));
        HIPCHK(hipEventCreate(&lane.t1));
        dmalloc(lane.small, dev::filter_small_words());
        zero_now(lane.small, 0, dev::filter_small_words() * sizeof(uint32_t)); // (the superblock counts start at zero; every sequence leaves them so)
        dmalloc(lane.d_scratch, (size_t)L_N);
        HIPCHK(hipHostMalloc((void**)&lane.h_scratch, L_N * sizeof(unsigned long long), hipHostMallocDefault));
        zero_now(lane.d_scratch, 0, L_N * sizeof(unsigned long long));
        lane.scratch_zero = true;
    }
    for (int j = 0; j < n; ++j) grow_lane(lanes_[j], cap);
}

// One filtered sequence for reads [lane.r0, lane.r1) of the batch, asynchronous on `stream`: kernels, then the lane's
// counters to its pinned mirror, then the counters cleared again behind the copy (so that the next batch starts with its
// first kernel instead of a memset).
void Mapper::launch_lane(Lane& lane, hipStream_t stream, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads,
    uint64_t n_bases, uint32_t* covg, uint32_t* prg_reads, const PackedInfo* pk)
{
    if (!lane.scratch_zero) HIPCHK(hipMemsetAsync(lane.d_scratch, 0, L_N * sizeof(unsigned long long), stream));
    lane.scratch_zero = false;
    if (!lane.cand_gp && dev::gathered_list_requested()) dmalloc(lane.cand_gp, lane.raw_capacity); // (grow_lane frees it with the rest)
    if (!lane.raw_grp && bloom0_wbits_ && dev::group_records_requested()) dmalloc(lane.raw_grp, lane.raw_capacity);
    dev::SketchArgs a = sketch_args(d_bases, d_offsets, n_reads, n_bases, pk);
    a.n_hits = &lane.d_scratch[L_HITS];
    a.n_minimizers = &lane.d_scratch[L_MINIMIZERS];
    a.overflow = reinterpret_cast<uint32_t*>(&lane.d_scratch[L_OVERFLOW]);
    dev::FilterBuffers fb { lane.raw_pos, lane.raw_grp, lane.cand_gp, lane.cand_info, lane.cand_pos1, lane.cand_rec, lane.raw_capacity, lane.small,
        &lane.d_scratch[L_MAXLEN] };
    fb.stat = d_ft_stat_;
    {
        static const bool adapt = [] { const char* e = std::getenv("DRPRG_FT_ADAPT"); return !e || std::atoi(e) != 0; }();
        ft_adapt_ = adapt;
        const int pi = pk ? 1 : 0;
        lane.ft_packed = pi != 0;
        if (adapt) {
            fb.class_clock = &lane.d_scratch[L_FT_CLOCK];
            if (ft_share_[pi][0]) fb.wave_share = ft_share_[pi];
        }
    }
    dev::BloomTables bt { d_bloom_, bloom_wbits_, d_bloom0_, bloom0_wbits_, d_bloomr_, d_bloom0f_ };
    bt.blkc = d_blkc_;
    bt.blkc_wbits = blkc_wbits_;
    if (use_mid_) {
        bt.mid0 = d_mid0_;
        bt.mid_bitmap = d_mid_bitmap_;
        bt.midc = d_midc_;
        bt.midc_wbits = midc_wbits_;
        bt.mid0_bits = mid0_bits_;
    }
    dev::ReadClusterArgs rc {};
    rc.prg_min_path_len = d_min_path_len_;
    rc.fraction = params_.cluster_fraction();
    rc.min_cluster_size = params_.min_cluster_size;
    rc.max_diff = params_.max_diff;
    rc.n_prgs = n_prgs_;
    rc.covg = covg;
    rc.prg_reads = prg_reads;
    rc.n_clusters_kept = &d_counters_[C_CLUSTERS_KEPT];
    rc.n_hits_kept = &d_counters_[C_HITS_KEPT];
    rc.n_complex = &lane.d_scratch[L_COMPLEX];
    rc.n_unfit = &lane.d_scratch[L_UNFIT];
    rc.chunk_flags = lane.rc_flags;
    rc.wg_partials = lane.rc_partials;
    rc.wg_done = lane.rc_flags + (lane.raw_capacity / dev::RC_CHUNK_OWN + 2); // (one word behind the flags, cleared with them)
    if (dev::read_cluster_wave_form_requested()) // (only the opt-in wave form reads them)
        HIPCHK(hipMemsetAsync(lane.rc_flags, 0, (lane.raw_capacity / dev::RC_CHUNK_OWN + 3) * sizeof(uint32_t), stream));
    rc.chunk_counter = reinterpret_cast<uint32_t*>(&lane.d_scratch[L_CHUNK]);
    dev::KernelTimer timer;
    if (timing_) { // events bracket the dominant kernel only
        timer.begin = lane.t0;
        timer.end = lane.t1;
    }
    HIPCHK(dev::launch_sketch_filter(a, lane.r0, lane.r1, bt, n_cus_, fb, rc, lane.fw, stream, timer));
    // the counters to the pinned mirror and zero again behind it: one small kernel (DRPRG_HIP_COUNTERS_HOME=0: a copy and a memset, rounds 1-4)
    static const bool one_launch = [] { const char* e = std::getenv("DRPRG_HIP_COUNTERS_HOME"); return !e || std::atoi(e) != 0; }();
    if (one_launch) {
        if (!lane.h_scratch_dev) HIPCHK(hipHostGetDevicePointer((void**)&lane.h_scratch_dev, lane.h_scratch, 0));
        HIPCHK(dev::launch_counters_home(lane.d_scratch, lane.h_scratch_dev, L_N, stream, dev::filter_super_counts(lane.small), dev::filter_super_words()));
    } else {
        HIPCHK(hipMemcpyAsync(lane.h_scratch, lane.d_scratch, L_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemsetAsync(lane.d_scratch, 0, L_N * sizeof(unsigned long long), stream));
        HIPCHK(hipMemsetAsync(dev::filter_super_counts(lane.small), 0, dev::filter_super_words() * sizeof(uint32_t), stream));
    }
    lane.scratch_zero = true;
}

// The wait polls the stream for a short while first: an interrupt-driven hipStreamSynchronize wakes up tens of microseconds late,
// which is visible at 0.7 ms per batch.  Only for a short while (~2 ms of polls): a longer batch does not notice the late wake-up,
// and a host core that spins for it is a core the FASTQ parser threads (or the other ranks of a node) do not have.
// DRPRG_HIP_SPIN=0: never poll.
void Mapper::wait_stream(hipStream_t stream)
{
    static const bool spin = [] {
        const char* e = std::getenv("DRPRG_HIP_SPIN");
        return !(e && std::atoi(e) == 0);
    }();
    for (int spins = 0; spin && spins < 2048; ++spins) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIPCHK(e);
    }
    HIPCHK(hipStreamSynchronize(stream));
}

// hits in d_key_a_/d_val_a_ -> clusters -> coverage (the generic pipeline).  ordered: the hits are ordered by
// (read, position) already and no read is longer than READ_SORT_MAX_LEN, so a per-read reorder replaces the radix sort.
void Mapper::cluster_hits(const uint64_t* d_offsets, uint32_t n_hits, bool ordered, unsigned long long* d_unsorted, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream)
{
    if (n_hits == 0) return;
    const uint64_t* s_key = d_key_b_;
    const uint32_t* s_val = d_val_b_;
    if (ordered) {
        HIPCHK(dev::launch_read_sort(d_key_a_, d_val_a_, n_hits, d_order_, hit_capacity_, d_unsorted, stream));
        s_key = d_key_a_;
        s_val = d_val_a_;
    } else HIPCHK(dev::sort_hits(d_temp_, temp_bytes_, d_key_a_, d_key_b_, d_val_a_, d_val_b_, n_hits, stream));
    HIPCHK(dev::launch_cluster_flags(s_key, n_hits, params_.max_diff, d_head_, d_scan_, d_temp_, temp_bytes_, stream));
    HIPCHK(dev::launch_cluster_starts(d_head_, d_scan_, n_hits, d_cstart_, stream));
    dev::ClusterArgs c {};
    c.key = s_key;
    c.val = s_val;
    c.scan = d_scan_;
    c.cstart = d_cstart_;
    c.d_n_clusters = d_scan_ + (n_hits - 1);
    c.offsets = d_offsets;
    c.prg_min_path_len = d_min_path_len_;
    c.clusters = d_clusters_;
    c.order = d_order_;
    c.w = params_.w;
    c.fraction = params_.cluster_fraction();
    c.min_cluster_size = params_.min_cluster_size;
    c.covg = covg;
    c.prg_reads = prg_reads;
    c.n_clusters_kept = &d_counters_[C_CLUSTERS_KEPT];
    c.n_hits_kept = &d_counters_[C_HITS_KEPT];
    HIPCHK(dev::launch_cluster_pipeline(c, n_hits, n_prgs_, stream));
}

dev::SketchArgs Mapper::sketch_args(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, const PackedInfo* pk) const
{
    dev::SketchArgs a {};
    a.bases = d_bases;
    a.offsets = d_offsets;
    a.n_bases = n_bases;
    a.n_reads = n_reads;
    a.w = params_.w;
    a.k = params_.k;
    a.halo = halo_;
    a.slot_key = d_slot_key_;
    a.slot_rec = d_slot_rec_;
    a.slot_first = d_slot_first_;
    a.table_bits = table_bits_;
    a.rec_knode = d_rec_knode_;
    a.rec_prg = d_rec_prg_;
    a.tile_first_read = ts_->d_tile_first;
    a.pbloom = d_pbloom_;
    a.pbloom_wbits = pbloom_wbits_;
    a.hit_key = d_key_a_;
    a.hit_val = d_val_a_;
    a.hit_capacity = hit_capacity_;
    a.n_hits = &d_counters_[C_HITS];
    a.n_minimizers = &d_counters_[C_MINIMIZERS];
    a.overflow = reinterpret_cast<uint32_t*>(&d_counters_[C_OVERFLOW]);
    if (pk) { // (a batch of 2-bit words: said by the caller, carried with the batch -- not looked up by its address)
        a.packed = 1;
        a.npos = pk->d_npos;
        a.n_npos = pk->n_npos;
    }
    return a;
}

const uint8_t* Mapper::ascii_view(int slot, const uint8_t* d_bases, uint64_t n_bases, hipStream_t stream, const PackedInfo* pk)
{
    if (!pk) return d_bases;
    const uint64_t need = (n_bases + 15) / 16 * 16 + 64;
    if (need > unpacked_cap_[slot]) {
        sync(); // (a batch in flight may still read the old buffer)
        HIPCHK(hipStreamSynchronize(stream));
        dfree(d_unpacked_[slot]);
        unpacked_cap_[slot] = need + need / 4;
        dmalloc(d_unpacked_[slot], unpacked_cap_[slot]);
    }
    HIPCHK(dev::launch_unpack(reinterpret_cast<const uint32_t*>(d_bases), n_bases, pk->d_npos, pk->n_npos, d_unpacked_[slot], stream));
    return d_unpacked_[slot];
}

void Mapper::read_counters(hipStream_t stream)
{
    HIPCHK(hipMemcpyAsync(h_counters_, d_counters_, C_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    wait_stream(stream);
}

void Mapper::note_kernel_time()
{
    if (!timing_) return;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    sketch_ms_ += ms;
    sketch_launches_ += 1;
}

// The reads read_cluster_kernel left over in this lane's candidate list: their hits -> the generic cluster pipeline.
void Mapper::leftovers(Lane& lane, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk)
{
    if (lane.h_scratch[L_COMPLEX] == 0) return;
    dev::SketchArgs a = sketch_args(d_bases, d_offsets, n_reads, n_bases, pk);
    a.n_hits = &lane.d_scratch[L_HITS];
    a.n_minimizers = &lane.d_scratch[L_MINIMIZERS]; // (the recount pass does not count minimizers again)
    a.overflow = reinterpret_cast<uint32_t*>(&lane.d_scratch[L_OVERFLOW]);
    lane.scratch_zero = false; // the leftover pass uses the lane's counters again
    HIPCHK(dev::launch_filter_recount(a, lane.fw, stream));
    HIPCHK(hipMemcpyAsync(lane.h_scratch, lane.d_scratch, L_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    wait_stream(stream);
    const uint64_t n_left = lane.h_scratch[L_HITS];
    if (n_left == 0) return;
    ensure_workspace(std::max<uint64_t>(1u << 20, n_left + n_left / 8));
    a.hit_key = d_key_a_; // the hit buffers may have moved
    a.hit_val = d_val_a_;
    a.hit_capacity = hit_capacity_;
    HIPCHK(dev::launch_filter_expand(a, lane.fw, stream));
    cluster_hits(d_offsets, (uint32_t)n_left, lane.h_scratch[L_MAXLEN] <= READ_SORT_MAX_LEN, &lane.d_scratch[L_UNSORTED], covg, prg_reads, stream);
}

void Mapper::free_tile_set(TileSet& t)
{
    dfree(t.d_tile_info); dfree(t.d_tile_pos1); dfree(t.d_tile_count); dfree(t.d_tile_hits); dfree(t.d_tile_nmin); dfree(t.d_tile_prefix); dfree(t.d_tile_fast);
    dfree(t.d_nbits);
    t.nbits_cap = 0;
    dfree(t.d_tile_rec); dfree(t.d_tile_first);
    if (t.d_tile_temp) (void)hipFree(t.d_tile_temp);
    t.d_tile_temp = nullptr;
    for (hipEvent_t* e : { &t.done, &t.t0, &t.t1 })
        if (*e) (void)hipEventDestroy(*e);
    t = TileSet();
}

void Mapper::ensure_tile_workspace(TileSet& t, uint32_t n_tiles, uint32_t tile_cap)
{
    if (n_tiles <= t.ws_tiles && tile_cap <= t.ws_cap) return;
    dfree(t.d_tile_info); dfree(t.d_tile_pos1); dfree(t.d_tile_count); dfree(t.d_tile_hits); dfree(t.d_tile_nmin); dfree(t.d_tile_prefix); dfree(t.d_tile_fast); dfree(t.d_tile_rec);
    if (t.d_tile_temp) (void)hipFree(t.d_tile_temp);
    t.d_tile_temp = nullptr;
    t.ws_tiles = std::max(t.ws_tiles, n_tiles + n_tiles / 8 + 32);
    t.ws_cap = std::max(t.ws_cap, tile_cap);
    const size_t n = (size_t)t.ws_tiles * t.ws_cap;
    dmalloc(t.d_tile_info, n); dmalloc(t.d_tile_pos1, n); dmalloc(t.d_tile_rec, n);
    dmalloc(t.d_tile_count, (size_t)t.ws_tiles + 1); dmalloc(t.d_tile_hits, (size_t)t.ws_tiles + 1); dmalloc(t.d_tile_nmin, (size_t)t.ws_tiles + 1); dmalloc(t.d_tile_prefix, (size_t)t.ws_tiles + 1); dmalloc(t.d_tile_fast, (size_t)t.ws_tiles + 1);
    t.tile_temp_bytes = dev::scan_temp_bytes(t.ws_tiles + 1);
    HIPCHK(hipMalloc(&t.d_tile_temp, t.tile_temp_bytes ? t.tile_temp_bytes : 1));
}

// Direct sequence, candidate form: every k-mer hashed (any k, any w, any index size); each tile leaves its index minimizers
// as candidate records in position order, a scan + gather makes the dense ordered list and read_cluster_kernel takes it from
// there -- no hit list, no radix sort, no cluster kernels for the reads that fit it.
// One attempt, asynchronous on `stream`: the launches, the lane's counters to their pinned mirror, the counters cleared behind the copy.
void Mapper::direct_launch(int set, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream, bool timed_by_set_events, const PackedInfo* pk)
{
    TileSet& t = tsets_[set];
    ts_ = &t;
    const uint32_t n_tiles = dev::direct_candidate_tiles(n_bases, halo_, params_.k, params_.w, wide_hash_); // = slices
    const uint32_t n_first = dev::direct_first_read_tiles(n_bases, halo_, params_.k, params_.w, wide_hash_);
    if (n_first > t.first_cap) { // first read of every tile
        dfree(t.d_tile_first);
        t.first_cap = n_first + n_first / 4 + 16;
        dmalloc(t.d_tile_first, (size_t)t.first_cap);
    }
    ensure_lanes(set + 1, 0); // (the lanes exist; only this set's lane may grow: the other one may belong to a batch in flight)
    Lane& lane = lanes_[(size_t)set];
    grow_lane(lane, std::min<uint64_t>(std::max<uint64_t>(1u << 20, n_bases / 16), (1ull << 31) - 1));
    // read_cluster_kernel takes the candidates straight from the tile slices (no gathered list unless reads are left over); what it
    // handled is marked in the dense cand_pos1 array with a value no other batch used (DRPRG_RC_SLICES=0: gather first, as before)
    static const bool from_slices = [] {
        const char* e = std::getenv("DRPRG_RC_SLICES");
        return !(e && std::atoi(e) == 0);
    }();
    uint32_t mark = 0;
    if (from_slices && !fuse_in_kernel_) {
        // (never a value a position + 1 can have, never 0; one sequence of epochs per process, so that no two Mappers that follow
        // each other in recycled device memory ever use the same mark)
        static std::atomic<uint32_t> process_epoch { 0x80000000u };
        slices_epoch_ = process_epoch.fetch_add(1) + 1;
        if (slices_epoch_ < 0x80000000u) { // wrapped after 2^31 batches: marks of old batches may still be around, clear them
            HIPCHK(hipMemsetAsync(lane.cand_pos1, 0, lane.raw_capacity * sizeof(uint32_t), stream));
            process_epoch.store(0x80000001u);
            slices_epoch_ = 0x80000001u;
        }
        mark = slices_epoch_;
    }
    ensure_tile_workspace(t, n_tiles, t.slice_cap);
    if (!lane.scratch_zero) HIPCHK(hipMemsetAsync(lane.d_scratch, 0, L_N * sizeof(unsigned long long), stream));
    lane.scratch_zero = false;
    HIPCHK(hipMemsetAsync(t.d_tile_count + n_tiles, 0, sizeof(uint32_t), stream)); // the scan's closing zero
    // a packed batch: sketch_wave_kernel reads the words themselves (round 4; the positions of its npos as one bit per base, set here);
    // the general direct kernel reads an ASCII expansion made on the same stream (packed.hip)
    const bool native_packed = pk != nullptr && dev::direct_uses_wave_form(params_.k, params_.w, wide_hash_);
    dev::SketchArgs a = sketch_args(native_packed ? d_bases : ascii_view(set, d_bases, n_bases, stream, pk), d_offsets, n_reads, n_bases, native_packed ? pk : nullptr);
    if (native_packed && a.n_npos) {
        const uint64_t words16 = ((n_bases + 15) / 16 + 1) & ~1ull; // (an even number of u16 words: the marks are 32-bit atomics)
        if (words16 > t.nbits_cap) {
            dfree(t.d_nbits);
            t.nbits_cap = words16 + words16 / 4;
            t.nbits_cap += t.nbits_cap & 1;
            dmalloc(t.d_nbits, (size_t)t.nbits_cap);
            t.nbits_dirty = true; // (fresh memory)
        }
        if (t.nbits_dirty) HIPCHK(hipMemsetAsync(t.d_nbits, 0, (size_t)t.nbits_cap * sizeof(uint16_t), stream));
        HIPCHK(dev::launch_mark_npos(a.npos, a.n_npos, n_bases, t.d_nbits, stream));
        t.nbits_dirty = true;
        a.nbits = t.d_nbits;
    }
    a.n_hits = &lane.d_scratch[L_HITS];
    a.n_minimizers = &lane.d_scratch[L_MINIMIZERS];
    a.overflow = reinterpret_cast<uint32_t*>(&lane.d_scratch[L_OVERFLOW]);
    a.tile_cap = t.ws_cap;
    a.tile_info = t.d_tile_info;
    a.tile_pos1 = t.d_tile_pos1;
    a.tile_rec = t.d_tile_rec;
    a.tile_count = t.d_tile_count;
    a.tile_hits = t.d_tile_hits;
    a.tile_nmin = t.d_tile_nmin;
    a.tile_fast = t.d_tile_fast;
    a.prg_min_path_len = d_min_path_len_;
    a.prg_thr = d_prg_thr_;
    // sketch_wave_kernel can cluster the reads that lie inside one tile itself and add their coverage on the spot
    // (opt-in: DRPRG_WAVE_FUSE=1; never with DRPRG_FT_DEBUG=8, "every read through the generic pipeline")
    a.fuse = fuse_in_kernel_ && dev::direct_uses_wave_form(params_.k, params_.w, wide_hash_) ? fuse_mode_ : 0;
    a.max_diff = params_.max_diff;
    a.covg = covg;
    a.prg_reads = prg_reads;
    a.n_clusters_kept = &d_counters_[C_CLUSTERS_KEPT];
    a.n_hits_kept = &d_counters_[C_HITS_KEPT];
    a.dbg = std::getenv("DRPRG_WAVE_DEBUG") ? &d_counters_[C_CHUNK] : nullptr; // (words C_CHUNK.. are unused by this sequence)
    a.fraction = params_.cluster_fraction();
    a.min_cluster_size = params_.min_cluster_size;
    dev::FilterBuffers fb { lane.raw_pos, lane.raw_grp, lane.cand_gp, lane.cand_info, lane.cand_pos1, lane.cand_rec, lane.raw_capacity, lane.small,
        &lane.d_scratch[L_MAXLEN] };
    dev::ReadClusterArgs rc {};
    rc.prg_min_path_len = d_min_path_len_;
    rc.fraction = params_.cluster_fraction();
    rc.min_cluster_size = params_.min_cluster_size;
    rc.max_diff = params_.max_diff;
    rc.n_prgs = n_prgs_;
    rc.covg = covg;
    rc.prg_reads = prg_reads;
    rc.n_clusters_kept = &d_counters_[C_CLUSTERS_KEPT];
    rc.n_hits_kept = &d_counters_[C_HITS_KEPT];
    rc.n_complex = &lane.d_scratch[L_COMPLEX];
    rc.n_unfit = &lane.d_scratch[L_UNFIT];
    rc.chunk_flags = lane.rc_flags;
    rc.wg_partials = lane.rc_partials;
    rc.wg_done = lane.rc_flags + (lane.raw_capacity / dev::RC_CHUNK_OWN + 2); // (one word behind the flags, cleared with them)
    if (dev::read_cluster_wave_form_requested()) // (only the opt-in wave form reads them)
        HIPCHK(hipMemsetAsync(lane.rc_flags, 0, (lane.raw_capacity / dev::RC_CHUNK_OWN + 3) * sizeof(uint32_t), stream));
    rc.chunk_counter = reinterpret_cast<uint32_t*>(&lane.d_scratch[L_CHUNK]);
    lane.fw = dev::FilterWork {};
    lane.fw.read_begin = 0;
    lane.fw.read_end = n_reads;
    dev::init_candidate_work(lane.fw, fb, n_cus_);
    dev::KernelTimer timer;
    if (timing_) {
        if (timed_by_set_events) {
            if (!t.t0) HIPCHK(hipEventCreate(&t.t0));
            if (!t.t1) HIPCHK(hipEventCreate(&t.t1));
            timer.begin = t.t0;
            timer.end = t.t1;
        } else {
            timer.begin = ev0_;
            timer.end = ev1_;
        }
    }
    HIPCHK(dev::launch_direct_candidates(a, wide_hash_, t.d_tile_prefix, t.d_tile_temp, t.tile_temp_bytes, lane.raw_capacity, rc, n_cus_, lane.fw,
        stream, timer, mark));
    t.a_done = a;
    t.mark = mark;
    t.n_tiles = n_tiles;
    HIPCHK(hipMemcpyAsync(lane.h_scratch, lane.d_scratch, L_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemsetAsync(lane.d_scratch, 0, L_N * sizeof(unsigned long long), stream));
    lane.scratch_zero = true;
}

// The read-back of such an attempt has arrived.  false: a tile slice or the dense list was too small -- nothing was counted except the
// minimizers, and those only in this attempt's scratch block --, the buffers have been grown and the caller runs the batch again.
// What sketch_wave_kernel already added to the coverage vector for the reads it clusters itself is taken back first: the same launch
// with fuse = -1 repeats exactly those additions as subtractions (same input, same slice capacity, so the same reads take that path).
// true: totals taken, reads left to the generic pipeline queued on `stream`.
bool Mapper::direct_finish(int set, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream, int attempt, const PackedInfo* pk)
{
    TileSet& t = tsets_[set];
    ts_ = &t;
    Lane& lane = lanes_[(size_t)set];
    const uint32_t ovf = (uint32_t)lane.h_scratch[L_OVERFLOW];
    if (ovf & 2u) throw Error(DRPRG_EOVERFLOW, "a read is longer than 2^" + std::to_string(dev::HIT_POS_BITS) + " bases");
    if (ovf & 4u) {
        if (attempt > 6) throw Error(DRPRG_EOVERFLOW, "candidate buffer overflow after regrow");
        if (t.a_done.fuse > 0) {
            dev::SketchArgs undo = t.a_done;
            undo.fuse = -t.a_done.fuse;
            HIPCHK(dev::launch_sketch_wave(undo, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        HIPCHK(hipStreamSynchronize(stream)); // (the buffers are about to be freed)
        t.slice_cap = std::min<uint32_t>(t.slice_cap * 2, 4096);
        grow_lane(lane, std::min<uint64_t>(lane.raw_capacity * 2, (1ull << 31) - 1));
        return false;
    }
    tot_minimizers_ += lane.h_scratch[L_MINIMIZERS];
    tot_hits_ += lane.h_scratch[L_HITS];
    tot_leftover_ += lane.h_scratch[L_COMPLEX];
    if (t.mark && lane.h_scratch[L_COMPLEX]) // reads were left over: the generic pipeline wants the gathered list after all
        HIPCHK(dev::launch_tile_gather_marked(t.a_done, lane.fw, t.d_tile_prefix, t.n_tiles, lane.raw_capacity, t.mark, stream));
    leftovers(lane, d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, pk);
    return true;
}

void Mapper::run_batch_direct_candidates(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases,
    uint32_t* covg, uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk)
{
    for (int attempt = 0;; ++attempt) {
        direct_launch(0, d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, false, pk);
        wait_stream(stream);
        note_kernel_time();
        if (direct_finish(0, d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, attempt, pk)) break;
    }
}

// What the host does with a lane's read-back once its sequence has finished: a candidate slice that was too small -> the range
// again with larger buffers; the totals; the reads read_cluster_kernel left over -> the generic pipeline on their hits.
void Mapper::finish_lane(Lane& lane, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk)
{
    for (int attempt = 0;; ++attempt) {
        const uint32_t ovf = (uint32_t)lane.h_scratch[L_OVERFLOW];
        if (ovf & 8u) throw Error(DRPRG_EIO, "sketch_filter_kernel: dynamic LDS does not start at address 0");
        if (ovf & 16u) throw Error(DRPRG_EIO, "sketch_filter_kernel: a chunk of its schedule holds too few whole tiles");
        if (ovf & 32u) throw Error(DRPRG_EINVAL, "the batch's read offsets do not span its bases: offsets[0] must be 0 and offsets[n_reads] must be n_bases");
        if (ovf & 2u) throw Error(DRPRG_EOVERFLOW, "a read is longer than 2^" + std::to_string(dev::HIT_POS_BITS) + " bases");
        if (!(ovf & 4u)) break;
        // a candidate slice of this range was too small: its sequence counted nothing and touched no coverage
        // (hit_scan_kernel / read_cluster_kernel check the flag); grow the lane and run the range again, alone
        if (attempt > 8) throw Error(DRPRG_EOVERFLOW, "candidate buffer overflow after regrow");
        grow_lane(lane, lane.raw_capacity * 4);
        launch_lane(lane, stream, d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, pk);
        wait_stream(stream);
    }
    tot_minimizers_ += lane.h_scratch[L_MINIMIZERS];
    tot_hits_ += lane.h_scratch[L_HITS];
    tot_leftover_ += lane.h_scratch[L_COMPLEX];
    tune_filter_shares(lane, lane.ft_packed, n_bases);
    leftovers(lane, d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, pk);
}

// sketch_filter_kernel's four wave classes should end together (sketch_filter.hip: a SIMD issues for its oldest wave first).  How far apart they
// ended in the batch just read back moves the next batch's shares: share_c *= (mean end / end_c)^0.6, every share kept within 0.35 .. 2.2 of an
// even one, the sum at 1024.  Any shares give the same candidates; batches too small to time (under 64 M bases) change nothing.
void Mapper::filter_schedule(uint64_t out[20])
{
    sync();
    for (int i = 0; i < 20; ++i) out[i] = ft_last_[i];
}

void Mapper::tune_filter_shares(const Lane& lane, bool packed, uint64_t n_bases)
{
    {   // what the batch just completed ran with (whatever its size)
        const dev::FilterSched& sc = lane.fw.sched;
        const unsigned long long* ck = &lane.h_scratch[L_FT_CLOCK];
        ft_last_[0] = sc.n_rounds;
        ft_last_[1] = lane.fw.n_slices;
        ft_last_[2] = sc.tpw0;
        for (int c = 0; c < 4; ++c) {
            ft_last_[3 + c] = lane.fw.wave_share[c];
            ft_last_[7 + c] = ck[0] && ck[1 + c] > ~ck[0] ? ck[1 + c] - ~ck[0] : 0;
        }
        ft_last_[11] = sc.per_wg;
        for (int r = 0; r < dev::FT_MAX_ROUNDS; ++r) ft_last_[12 + r] = r && r < (int)sc.n_rounds ? sc.size[r] : 0;
    }
    if (lane.fw.cand_total && std::getenv("DRPRG_FT_STATS")) { // what the filter left for verify_scan_kernel (measurements; the batch is complete)
        uint32_t total = 0;
        if (hipMemcpy(&total, lane.fw.cand_total, sizeof(total), hipMemcpyDeviceToHost) == hipSuccess)
            std::fprintf(stderr, "[sketch_filter] candidate positions of the batch: %u\n", total);
    }
    if (lane.fw.sched.n_rounds > 1) return; // a dynamic schedule balances itself: the shares of round 0 stay what they are
    if (!ft_adapt_ || n_bases < (64ull << 20)) return;
    if (max_lanes_ > 1) return; // (read ranges on concurrent streams: the classes' clocks measure the overlap, not the shares -- ADVICE r05)
    const unsigned long long* ck = &lane.h_scratch[L_FT_CLOCK];
    if (!ck[0] || !ck[1] || !ck[2] || !ck[3] || !ck[4]) return; // (a launch without the level-0 form, or with DRPRG_FT_SHARE)
    const unsigned long long t0 = ~ck[0];
    double end[4], mean = 0;
    for (int c = 0; c < 4; ++c) {
        if (ck[1 + c] <= t0) return;
        end[c] = (double)(ck[1 + c] - t0);
        mean += end[c] / 4;
    }
    uint32_t* s = ft_share_[packed ? 1 : 0];
    if (!s[0]) { // the launcher's built-in shares were in use: the same numbers (sketch_filter.hip)
        static const uint32_t ascii_l0[4] = { 397, 294, 200, 133 }, packed_l0[4] = { 422, 292, 184, 126 }, mid_l0[4] = { 356, 292, 220, 156 };
        const uint32_t* from = use_mid_ ? mid_l0 : packed ? packed_l0 : ascii_l0;
        for (int c = 0; c < 4; ++c) s[c] = from[c];
    }
    double v[4], sum = 0;
    for (int c = 0; c < 4; ++c) {
        v[c] = (double)s[c] * std::pow(mean / end[c], 0.6);
        v[c] = std::min(std::max(v[c], 0.35 * 256), 2.2 * 256);
        sum += v[c];
    }
    uint32_t acc = 0;
    for (int c = 0; c < 3; ++c) acc += s[c] = (uint32_t)(1024.0 * v[c] / sum + 0.5);
    s[3] = 1024u - acc;
    static const bool verbose = [] { const char* e = std::getenv("DRPRG_FT_ADAPT"); return e && std::atoi(e) == 2; }();
    if (verbose) std::fprintf(stderr, "[ft shares] classes ended at %.0f %.0f %.0f %.0f (10 ns) -> %u %u %u %u\n", end[0], end[1], end[2], end[3], s[0], s[1], s[2], s[3]);
}

void Mapper::map_device_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream)
{
    map_device_async_impl(d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, nullptr);
}

void Mapper::map_device_async_impl(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t* covg,
    uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk)
{
    if (n_reads == 0) return;
    const bool deferred_filter = use_filter_ && max_lanes_ == 1;
    const bool deferred_direct = !use_filter_ && use_direct_cands_ && !fuse_in_kernel_;
    if ((!deferred_filter && !deferred_direct) || n_bases == 0) { // (no deferred form of the other sequences: the batch is complete on return,
        map_device_impl(d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, pk); // including what its tail queued for the leftover reads)
        HIPCHK(hipSetDevice(device_));
        wait_stream(stream ? stream : stream_);
        return;
    }
    if (!d_bases || !d_offsets) throw Error(DRPRG_EINVAL, "null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_bases) & 15u) != 0) throw Error(DRPRG_EINVAL, "d_bases must be 16-byte aligned");
    if (n_reads > dev::MAX_BATCH_READS)
        throw Error(DRPRG_EOVERFLOW, "at most " + std::to_string(dev::MAX_BATCH_READS) + " reads per batch");
    HIPCHK(hipSetDevice(device_));
    if (kept_cap_ && !in_keep_call_) kept_broken_ = true;
    if (!stream) stream = stream_;
    if (!covg) covg = d_covg_;
    if (!prg_reads) prg_reads = d_prg_reads_;
    if (deferred_direct) {
        // the direct sequence in its candidate form, deferred the same way: two sets of the tile workspace taken in turn, the batch's
        // read-back looked at while the next batch runs (a batch whose slices overflowed, or that left reads to the generic pipeline,
        // is finished then -- from its own set, which the batch in flight does not touch)
        const int set = pipe_next_;
        direct_launch(set, d_bases, d_offsets, (uint32_t)n_reads, n_bases, covg, prg_reads, stream, true, pk);
        TileSet& t = tsets_[set];
        if (!t.done) HIPCHK(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
        HIPCHK(hipEventRecord(t.done, stream));
        Pending cur;
        cur.active = true;
        cur.direct = true;
        cur.lane = set;
        cur.packed = pk != nullptr;
        if (pk) cur.pk = *pk;
        cur.d_bases = d_bases;
        cur.d_offsets = d_offsets;
        cur.n_reads = (uint32_t)n_reads;
        cur.n_bases = n_bases;
        cur.covg = covg;
        cur.prg_reads = prg_reads;
        cur.stream = stream;
        const Pending prev = pending_;
        pending_ = cur;
        pipe_next_ ^= 1;
        tot_reads_ += n_reads;
        tot_bases_ += n_bases;
        if (prev.active) complete_batch(prev);
        return;
    }
    if (pipe_lanes_.empty()) {
        for (int j = 0; j < 2; ++j) {
            pipe_lanes_.emplace_back();
            Lane& lane = pipe_lanes_.back();
            HIPCHK(hipEventCreateWithFlags(&lane.done, hipEventDisableTiming));
            HIPCHK(hipEventCreate(&lane.t0));
            HIPCHK(hipEventCreate(&lane.t1));
            dmalloc(lane.small, dev::filter_small_words());
            zero_now(lane.small, 0, dev::filter_small_words() * sizeof(uint32_t));
            dmalloc(lane.d_scratch, (size_t)L_N);
            HIPCHK(hipHostMalloc((void**)&lane.h_scratch, L_N * sizeof(unsigned long long), hipHostMallocDefault));
            zero_now(lane.d_scratch, 0, L_N * sizeof(unsigned long long));
            lane.scratch_zero = true;
        }
    }
    Lane& lane = pipe_lanes_[(size_t)pipe_next_];
    // (this lane's previous batch was completed by the call before this one; growing frees its buffers, which waits for the device)
    grow_lane(lane, std::max<uint64_t>(1u << 20, n_bases / 64));
    lane.r0 = 0;
    lane.r1 = (uint32_t)n_reads;
    launch_lane(lane, stream, d_bases, d_offsets, (uint32_t)n_reads, n_bases, covg, prg_reads, pk);
    HIPCHK(hipEventRecord(lane.done, stream));
    Pending cur;
    cur.active = true;
    cur.lane = pipe_next_;
    cur.packed = pk != nullptr;
    if (pk) cur.pk = *pk;
    cur.d_bases = d_bases;
    cur.d_offsets = d_offsets;
    cur.n_reads = (uint32_t)n_reads;
    cur.n_bases = n_bases;
    cur.covg = covg;
    cur.prg_reads = prg_reads;
    cur.stream = stream;
    // the batch just queued is registered before the previous one is completed: if completing that one throws (a read too long,
    // an overflow that does not go away, a HIP error) the queued batch is still known to sync() and to the next call
    const Pending prev = pending_;
    pending_ = cur;
    pipe_next_ ^= 1;
    tot_reads_ += n_reads;
    tot_bases_ += n_bases;
    if (prev.active) complete_batch(prev); // the batch before this one, while this one runs
}

void Mapper::complete_pending()
{
    if (!pending_.active) return;
    const Pending p = pending_;
    pending_.active = false;
    complete_batch(p);
}

void Mapper::complete_batch(const Pending& p)
{
    HIPCHK(hipSetDevice(device_));
    const PackedInfo* const ppk = p.packed ? &p.pk : nullptr; // (the batch's own description: a re-run or its leftover reads see the format it came in)
    if (p.direct) {
        TileSet& t = tsets_[p.lane];
        static const bool spin_d = [] {
            const char* e = std::getenv("DRPRG_HIP_SPIN");
            return !(e && std::atoi(e) == 0);
        }();
        bool ready = false;
        for (int spins = 0; spin_d && spins < 2048 && !ready; ++spins) {
            const hipError_t e = hipEventQuery(t.done);
            if (e == hipSuccess) ready = true;
            else if (e != hipErrorNotReady) HIPCHK(e);
        }
        if (!ready) HIPCHK(hipEventSynchronize(t.done));
        if (timing_ && t.t0 && t.t1) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.t0, t.t1) == hipSuccess) {
                sketch_ms_ += ms;
                sketch_launches_ += 1;
            }
        }
        const uint64_t leftover_before = tot_leftover_;
        bool reran = false;
        for (int attempt = 0; !direct_finish(p.lane, p.d_bases, p.d_offsets, p.n_reads, p.n_bases, p.covg, p.prg_reads, p.stream, attempt, ppk); ++attempt) {
            // its slices were too small: nothing of it was counted; again, alone, behind whatever is queued on its stream
            direct_launch(p.lane, p.d_bases, p.d_offsets, p.n_reads, p.n_bases, p.covg, p.prg_reads, p.stream, false, ppk);
            wait_stream(p.stream);
            reran = true;
        }
        if (reran || tot_leftover_ != leftover_before) wait_stream(p.stream);
        return;
    }
    Lane& lane = pipe_lanes_[(size_t)p.lane];
    static const bool spin = [] {
        const char* e = std::getenv("DRPRG_HIP_SPIN");
        return !(e && std::atoi(e) == 0);
    }();
    bool ready = false;
    for (int spins = 0; spin && spins < 2048 && !ready; ++spins) {
        const hipError_t e = hipEventQuery(lane.done);
        if (e == hipSuccess) ready = true;
        else if (e != hipErrorNotReady) HIPCHK(e);
    }
    if (!ready) HIPCHK(hipEventSynchronize(lane.done));
    if (timing_) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, lane.t0, lane.t1));
        sketch_ms_ += ms;
        sketch_launches_ += 1;
    }
    // a batch that needed the host afterwards (run again with larger buffers, reads left to the generic pipeline) has just had
    // more work queued for its accumulators: "completed" means that work is done too -- the caller may touch the buffers of a
    // completed batch from any stream
    const uint64_t leftover_before = tot_leftover_;
    const bool overflowed = ((uint32_t)lane.h_scratch[L_OVERFLOW] & 4u) != 0;
    finish_lane(lane, p.d_bases, p.d_offsets, p.n_reads, p.n_bases, p.covg, p.prg_reads, p.stream, ppk);
    if (overflowed || tot_leftover_ != leftover_before) wait_stream(p.stream);
}

void Mapper::sync()
{
    const hipStream_t last = pending_.active ? pending_.stream : nullptr;
    complete_pending();
    if (last) HIPCHK(hipStreamSynchronize(last));
}

void Mapper::run_batch(const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_reads, uint64_t n_bases,
    uint32_t* covg, uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk)
{
    if (n_bases == 0) return; // only empty reads: no k-mers, no hits
    dev::KernelTimer timer;
    if (timing_) { // events bracket the dominant kernel only (sketch_filter_kernel / sketch_probe_kernel)
        timer.begin = ev0_;
        timer.end = ev1_;
    }
    if (use_filter_) {
        // ---- filtered sequences: the hits of short reads never leave the chip (read_cluster_kernel).  The batch is cut into
        // one read range per lane; the ranges run concurrently (lane 0 on the caller's stream), one host wait at the end ----
        const int n_lanes = (n_bases >= lanes_min_bases_ && n_reads >= 64) ? max_lanes_ : 1;
        const uint64_t lane_cap = std::max<uint64_t>(1u << 20, n_bases / 64 / (uint64_t)n_lanes * (n_lanes > 1 ? 3 : 2) / 2);
        ensure_lanes(n_lanes, lane_cap);
        if (n_lanes > 1) HIPCHK(hipEventRecord(ev_begin_, stream));
        for (int j = n_lanes - 1; j >= 0; --j) { // (lane 0 last: its stream is the one the host then waits on)
            Lane& lane = lanes_[j];
            lane.r0 = (uint32_t)((uint64_t)n_reads * (uint64_t)j / (uint64_t)n_lanes);
            lane.r1 = (uint32_t)((uint64_t)n_reads * (uint64_t)(j + 1) / (uint64_t)n_lanes);
            hipStream_t ls = j == 0 ? stream : lane.stream;
            if (j > 0) HIPCHK(hipStreamWaitEvent(ls, ev_begin_, 0));
            launch_lane(lane, ls, d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, pk);
            if (j > 0) HIPCHK(hipEventRecord(lane.done, ls));
        }
        for (int j = 1; j < n_lanes; ++j) HIPCHK(hipStreamWaitEvent(stream, lanes_[j].done, 0));
        wait_stream(stream);
        if (timing_) {
            for (int j = 0; j < n_lanes; ++j) {
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, lanes_[j].t0, lanes_[j].t1));
                sketch_ms_ += ms;
                sketch_launches_ += 1;
            }
        }
        for (int j = 0; j < n_lanes; ++j) finish_lane(lanes_[j], d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, pk);
        return;
    }
    if (use_direct_cands_) {
        run_batch_direct_candidates(d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, pk);
        return;
    }
    // ---- direct sequence, generic form: every k-mer hashed, hits in tile order, global radix sort ----
    ensure_workspace(std::max<uint64_t>(1u << 20, n_bases / 64));
    const uint32_t n_tiles = dev::sketch_n_tiles(n_bases, halo_);
    ts_ = &tsets_[0];
    if (n_tiles > ts_->first_cap) { // first read of every tile
        dfree(ts_->d_tile_first);
        ts_->first_cap = n_tiles + n_tiles / 4 + 16;
        dmalloc(ts_->d_tile_first, (size_t)ts_->first_cap);
    }
    const uint8_t* const ascii = ascii_view(2, d_bases, n_bases, stream, pk); // (a packed batch: expanded on the device for this kernel)
    for (int attempt = 0;; ++attempt) {
        HIPCHK(hipMemsetAsync(&d_counters_[C_HITS], 0, 2 * sizeof(unsigned long long), stream)); // hits + minimizers of this attempt
        HIPCHK(hipMemsetAsync(&d_counters_[C_OVERFLOW], 0, sizeof(unsigned long long), stream));
        const dev::SketchArgs a = sketch_args(ascii, d_offsets, n_reads, n_bases, nullptr); // (the expansion is ASCII)
        HIPCHK(dev::launch_sketch_probe(a, wide_hash_, stream, timer));
        read_counters(stream);
        note_kernel_time();
        if ((uint32_t)h_counters_[C_OVERFLOW] & 2u)
            throw Error(DRPRG_EOVERFLOW, "a read is longer than 2^" + std::to_string(dev::HIT_POS_BITS) + " bases");
        if (h_counters_[C_HITS] > hit_capacity_) {
            if (attempt > 6) throw Error(DRPRG_EOVERFLOW, "hit buffer overflow after regrow");
            ensure_workspace(h_counters_[C_HITS] + h_counters_[C_HITS] / 8 + 1024);
            continue;
        }
        break;
    }
    tot_minimizers_ += h_counters_[C_MINIMIZERS];
    tot_hits_ += h_counters_[C_HITS];
    cluster_hits(d_offsets, (uint32_t)h_counters_[C_HITS], false, nullptr, covg, prg_reads, stream);
}

void Mapper::map_device(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
    uint32_t* covg, uint32_t* prg_reads, hipStream_t stream)
{
    map_device_impl(d_bases, d_offsets, n_reads, n_bases, covg, prg_reads, stream, nullptr);
}

void Mapper::map_device_packed(const uint32_t* d_words, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases, const uint64_t* d_npos, uint64_t n_npos,
    uint32_t* covg, uint32_t* prg_reads, hipStream_t stream, bool deferred)
{
    if (n_npos && !d_npos) throw Error(DRPRG_EINVAL, "n_npos > 0 without the positions");
    if (n_reads == 0) return;
    // (everything that can refuse the batch is checked before the batch is known as packed; a batch that fails later is forgotten again)
    if (!d_words || !d_offsets) throw Error(DRPRG_EINVAL, "null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_words) & 15u) != 0) throw Error(DRPRG_EINVAL, "d_words must be 16-byte aligned");
    const uint8_t* key = reinterpret_cast<const uint8_t*>(d_words);
    const PackedInfo info { d_npos, n_npos }; // (travels with the batch through every function that touches it, and into Pending)
    if (deferred) map_device_async_impl(key, d_offsets, n_reads, n_bases, covg, prg_reads, stream, &info);
    else map_device_impl(key, d_offsets, n_reads, n_bases, covg, prg_reads, stream, &info);
}

void Mapper::map_device_impl(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
    uint32_t* covg, uint32_t* prg_reads, hipStream_t stream, const PackedInfo* pk)
{
    if (n_reads == 0) return;
    if (!d_bases || !d_offsets) throw Error(DRPRG_EINVAL, "null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_bases) & 15u) != 0) throw Error(DRPRG_EINVAL, "d_bases must be 16-byte aligned");
    if (n_reads > dev::MAX_BATCH_READS)
        throw Error(DRPRG_EOVERFLOW, "at most " + std::to_string(dev::MAX_BATCH_READS) + " reads per batch");
    HIPCHK(hipSetDevice(device_));
    if (kept_cap_ && !in_keep_call_) kept_broken_ = true; // reads of the caller's own buffers: not among the kept ones
    complete_pending();
    run_batch(d_bases, d_offsets, (uint32_t)n_reads, n_bases, covg ? covg : d_covg_, prg_reads ? prg_reads : d_prg_reads_,
        stream ? stream : stream_, pk);
    tot_reads_ += n_reads;
    tot_bases_ += n_bases;
}

void Mapper::map_host(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads)
{
    HostBatch b;
    b.bases = bases;
    b.offsets = offsets;
    b.n_reads = n_reads;
    map_host(b);
}

void Mapper::map_host(const HostBatch& b)
{
    const uint64_t n_reads = b.n_reads;
    if (n_reads == 0) return;
    sync();
    HIPCHK(hipSetDevice(device_));
    if (b.offsets[0] != 0) throw Error(DRPRG_EINVAL, "offsets[0] must be 0");
    const uint64_t n_bases = b.n_bases(), bytes = b.payload_bytes();
    if (bytes + 64 > stage_bases_cap_) {
        dfree(d_bases_);
        stage_bases_cap_ = bytes + bytes / 4 + 64;
        dmalloc(d_bases_, stage_bases_cap_);
    }
    if (n_reads + 1 > stage_reads_cap_) {
        dfree(d_offsets_);
        stage_reads_cap_ = n_reads + n_reads / 4 + 1;
        dmalloc(d_offsets_, stage_reads_cap_);
    }
    if (b.packed && b.n_npos > stage_npos_cap_) {
        dfree(d_npos_);
        stage_npos_cap_ = b.n_npos + b.n_npos / 4 + 16;
        dmalloc(d_npos_, stage_npos_cap_);
    }
    HIPCHK(hipMemcpyAsync(d_bases_, b.bases, bytes, hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(d_offsets_, b.offsets, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
    if (b.packed) {
        if (b.n_npos) HIPCHK(hipMemcpyAsync(d_npos_, b.npos, b.n_npos * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
        map_device_packed(reinterpret_cast<const uint32_t*>(d_bases_), d_offsets_, n_reads, n_bases, b.n_npos ? d_npos_ : nullptr, b.n_npos, nullptr, nullptr, stream_,
            false);
    } else {
        map_device(d_bases_, d_offsets_, n_reads, n_bases, nullptr, nullptr, stream_);
    }
    HIPCHK(hipStreamSynchronize(stream_)); // the staging buffers are reused by the next call
}

uint64_t Mapper::pack_on_device(const uint8_t* d_bases, uint64_t n_bases, uint32_t* d_words, uint64_t* d_npos, uint64_t npos_cap, hipStream_t stream)
{
    HIPCHK(hipSetDevice(device_));
    if (!stream) stream = stream_;
    if (n_bases == 0) return 0;
    if (!d_bases || !d_words) throw Error(DRPRG_EINVAL, "null device pointer");
    if (!d_pack_count_) dmalloc(d_pack_count_, (size_t)1); // (one counter word per Mapper: a hipMalloc / hipFree per call synchronises the device)
    unsigned long long n = 0;
    HIPCHK(hipMemsetAsync(d_pack_count_, 0, sizeof(unsigned long long), stream));
    HIPCHK(dev::launch_pack(d_bases, n_bases, d_words, d_npos, d_npos ? npos_cap : 0, d_pack_count_, stream));
    HIPCHK(hipMemcpyAsync(&n, d_pack_count_, sizeof n, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (n && !d_npos) throw Error(DRPRG_EINVAL, "the batch holds bases that are not ACGT and no buffer for their positions was given");
    if (d_npos && n > 1 && n <= npos_cap) { // the positions come out in any order: sorted on the host (few; the harness path)
        std::vector<uint64_t> h(n);
        HIPCHK(hipMemcpy(h.data(), d_npos, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        HIPCHK(hipMemcpy(d_npos, h.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    return n;
}

void* Mapper::arena_take(size_t bytes)
{
    bytes = (bytes + 255) / 256 * 256;
    if (kept_bytes_ + bytes > kept_cap_) return nullptr;
    if (bytes > arena_left_) {
        // (what is left of the current piece is not used: pieces are 256 MB, blocks ~25 MB)
        const size_t piece = std::max<size_t>(bytes, std::min<uint64_t>(256ull << 20, kept_cap_ - kept_bytes_));
        void* p = nullptr;
        if (hipMalloc(&p, piece) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        kept_arenas_.emplace_back(p, piece);
        arena_at_ = static_cast<uint8_t*>(p);
        arena_left_ = piece;
    }
    void* r = arena_at_;
    arena_at_ += bytes;
    arena_left_ -= bytes;
    kept_bytes_ += bytes;
    return r;
}

void Mapper::keep_reads(uint64_t max_bytes)
{
    drop_kept();
    kept_cap_ = max_bytes;
    // reads mapped before this call are not resident: kept_complete() must not claim them (map_resident would map a subset)
    kept_broken_ = tot_reads_ != 0;
}

void Mapper::drop_kept()
{
    if (!kept_arenas_.empty()) {
        sync();
        HIPCHK(hipSetDevice(device_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (auto& a : kept_arenas_) (void)hipFree(a.first);
    }
    kept_arenas_.clear();
    kept_.clear();
    arena_at_ = nullptr;
    arena_left_ = 0;
    kept_bytes_ = 0;
    kept_broken_ = false;
}

uint64_t Mapper::map_kept_from(const Mapper& other)
{
    if (other.device_ != device_) throw Error(DRPRG_EINVAL, "the kept reads live on another device");
    if (!other.kept_complete()) throw Error(DRPRG_ENODATA, "the other context does not hold all of its reads");
    uint64_t n = 0;
    for (const KeptBatch& b : other.kept_) {
        if (b.packed) map_device_packed(reinterpret_cast<const uint32_t*>(b.d_bases), b.d_offsets, b.n_reads, b.n_bases, b.d_npos, b.n_npos, nullptr, nullptr, stream_, true);
        else map_device_async(b.d_bases, b.d_offsets, b.n_reads, b.n_bases, nullptr, nullptr, stream_);
        n += b.n_reads;
    }
    sync();
    HIPCHK(hipStreamSynchronize(stream_));
    return n;
}

void Mapper::select_reads_with_anchors(std::vector<uint64_t> anchors, uint32_t A, std::vector<uint8_t>& bases, std::vector<uint64_t>& offsets)
{
    if (!kept_complete()) throw Error(DRPRG_ENODATA, "not every read of the sample is resident");
    if (A == 0 || A > 31) throw Error(DRPRG_EINVAL, "anchor length must be 1..31");
    if (offsets.empty()) offsets.push_back(0);
    std::sort(anchors.begin(), anchors.end());
    anchors.erase(std::unique(anchors.begin(), anchors.end()), anchors.end());
    if (anchors.empty() || kept_.empty()) return;
    sync();
    HIPCHK(hipSetDevice(device_));
    std::vector<uint32_t> pf(2048, 0);
    for (uint64_t a : anchors) pf[(a & 0xFFFF) >> 5] |= 1u << (a & 31);
    uint64_t total_reads = 0, max_reads = 0;
    for (const KeptBatch& b : kept_) {
        total_reads += b.n_reads;
        max_reads = std::max(max_reads, b.n_reads);
    }
    // device scratch of this call (freed on every way out)
    struct Scratch {
        std::vector<void*> p;
        ~Scratch()
        {
            for (void* q : p) (void)hipFree(q);
        }
        void* get(size_t bytes)
        {
            void* q = nullptr;
            if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) throw Error(DRPRG_ENOMEM, "out of device memory (read selection)");
            p.push_back(q);
            return q;
        }
    } scratch;
    uint64_t* d_anchors = static_cast<uint64_t*>(scratch.get(anchors.size() * sizeof(uint64_t)));
    uint32_t* d_pf = static_cast<uint32_t*>(scratch.get(pf.size() * sizeof(uint32_t)));
    uint32_t* d_flags = static_cast<uint32_t*>(scratch.get(max_reads * sizeof(uint32_t)));
    unsigned long long* d_count = static_cast<unsigned long long*>(scratch.get(sizeof(unsigned long long)));
    dev::SelectedRead* d_list = static_cast<dev::SelectedRead*>(scratch.get(total_reads * sizeof(dev::SelectedRead)));
    HIPCHK(hipMemcpyAsync(d_anchors, anchors.data(), anchors.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(d_pf, pf.data(), pf.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream_));
    // Batches kept in the packed form: the scan and the gather read an ASCII expansion.  It is made a WINDOW of batches at a time -- as
    // many consecutive batches as expand into <= 1 GB (a larger batch alone) --, scanned, and its selected reads gathered before the next
    // window's expansion takes the scratch: the scratch is the largest window, not the sample (up to 128 Gbases can stay
    // resident packed; expanding all of them at once needed that much memory again, with no way back to the file pass).
    auto expanded = [](const KeptBatch& kb) -> uint64_t { return kb.packed ? (kb.n_bases + 15) / 16 * 16 + 64 : 0; };
    constexpr uint64_t WINDOW_BYTES = 1ull << 30;
    std::vector<std::pair<size_t, size_t>> windows; // [first batch, one past the last)
    uint64_t scratch_need = 0;
    for (size_t b = 0; b < kept_.size();) {
        size_t e = b;
        uint64_t need = 0;
        do need += expanded(kept_[e++]);
        while (e < kept_.size() && need + expanded(kept_[e]) <= WINDOW_BYTES);
        windows.emplace_back(b, e);
        scratch_need = std::max(scratch_need, need);
        b = e;
    }
    uint8_t* d_ascii = scratch_need ? static_cast<uint8_t*>(scratch.get(scratch_need)) : nullptr;
    std::vector<const uint8_t*> ascii(kept_.size(), nullptr);
    unsigned long long done = 0; // entries of d_list the windows before this one left
    for (const auto& win : windows) {
        uint64_t at = 0;
        for (size_t b = win.first; b < win.second; ++b) {
            const KeptBatch& kb = kept_[b];
            if (!kb.packed) {
                ascii[b] = kb.d_bases;
                continue;
            }
            HIPCHK(dev::launch_unpack(reinterpret_cast<const uint32_t*>(kb.d_bases), kb.n_bases, kb.d_npos, kb.n_npos, d_ascii + at, stream_));
            ascii[b] = d_ascii + at;
            at += expanded(kb);
        }
        for (size_t b = win.first; b < win.second; ++b) {
            const KeptBatch& kb = kept_[b];
            HIPCHK(hipMemsetAsync(d_flags, 0, kb.n_reads * sizeof(uint32_t), stream_));
            HIPCHK(dev::launch_anchor_scan(ascii[b], kb.d_offsets, (uint32_t)kb.n_reads, kb.n_bases, d_anchors, (uint32_t)anchors.size(), A, d_pf, (uint32_t)b,
                d_flags, d_count, d_list, total_reads, n_cus_, stream_));
        }
        unsigned long long count = 0;
        HIPCHK(hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (count > total_reads) throw Error(DRPRG_EIO, "read selection: more reads than the batches hold"); // (each read is appended once)
        if (count == done) continue;
        std::vector<dev::SelectedRead> list(count - done);
        HIPCHK(hipMemcpy(list.data(), d_list + done, (count - done) * sizeof(dev::SelectedRead), hipMemcpyDeviceToHost));
        done = count;
        std::sort(list.begin(), list.end(), [](const dev::SelectedRead& x, const dev::SelectedRead& y) { return x.batch != y.batch ? x.batch < y.batch : x.read < y.read; });
        std::vector<dev::GatherEntry> table(list.size());
        uint64_t out_bytes = 0;
        for (size_t i = 0; i < list.size(); ++i) {
            table[i].src = ascii[list[i].batch] + list[i].offset;
            table[i].dst = out_bytes;
            table[i].len = list[i].len;
            table[i].pad = 0;
            out_bytes += list[i].len;
        }
        // (device memory of this window only: freed before the next one)
        dev::GatherEntry* d_table = nullptr;
        uint8_t* d_out = nullptr;
        if (hipMalloc((void**)&d_table, list.size() * sizeof(dev::GatherEntry)) != hipSuccess || hipMalloc((void**)&d_out, out_bytes ? out_bytes : 16) != hipSuccess) {
            if (d_table) (void)hipFree(d_table);
            throw Error(DRPRG_ENOMEM, "out of device memory (read selection)");
        }
        const size_t base0 = bases.size();
        bases.resize(base0 + out_bytes);
        hipError_t e = hipMemcpyAsync(d_table, table.data(), list.size() * sizeof(dev::GatherEntry), hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess) e = dev::launch_gather_reads(d_table, (uint32_t)list.size(), d_out, stream_);
        if (e == hipSuccess && out_bytes) e = hipMemcpyAsync(bases.data() + base0, d_out, out_bytes, hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        (void)hipFree(d_table);
        (void)hipFree(d_out);
        HIPCHK(e);
        for (size_t i = 0; i < list.size(); ++i) offsets.push_back(offsets.back() + list[i].len);
    }
}

void Mapper::map_host_async(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads)
{
    HostBatch b;
    b.bases = bases;
    b.offsets = offsets;
    b.n_reads = n_reads;
    map_host_async(b);
}

void Mapper::map_host_async(const HostBatch& hb)
{
    const uint64_t n_reads = hb.n_reads;
    const uint64_t* offsets = hb.offsets;
    if (n_reads == 0) return;
    if (kept_cap_ && !kept_broken_) {
        // the block goes into device memory of its own and stays there
        HIPCHK(hipSetDevice(device_));
        if (offsets[0] != 0) throw Error(DRPRG_EINVAL, "offsets[0] must be 0");
        const uint64_t n_bases = hb.n_bases(), bytes = hb.payload_bytes();
        uint8_t* db = n_bases ? static_cast<uint8_t*>(arena_take(bytes + 64)) : nullptr;
        uint64_t* doff = n_bases && db ? static_cast<uint64_t*>(arena_take((n_reads + 1) * sizeof(uint64_t))) : nullptr;
        uint64_t* dnp = n_bases && doff && hb.packed && hb.n_npos ? static_cast<uint64_t*>(arena_take(hb.n_npos * sizeof(uint64_t))) : nullptr;
        if (n_bases == 0) {
            tot_reads_ += n_reads; // (only empty reads: nothing to keep)
            return;
        }
        if (db && doff && (dnp || !(hb.packed && hb.n_npos))) {
            if (!copy_stream_) HIPCHK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
            if (!kept_copied_) HIPCHK(hipEventCreateWithFlags(&kept_copied_, hipEventDisableTiming));
            HIPCHK(hipMemcpyAsync(db, hb.bases, bytes, hipMemcpyHostToDevice, copy_stream_));
            HIPCHK(hipMemcpyAsync(doff, offsets, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, copy_stream_));
            if (dnp) HIPCHK(hipMemcpyAsync(dnp, hb.npos, hb.n_npos * sizeof(uint64_t), hipMemcpyHostToDevice, copy_stream_));
            HIPCHK(hipEventRecord(kept_copied_, copy_stream_));
            HIPCHK(hipStreamWaitEvent(stream_, kept_copied_, 0));
            KeptBatch kb { db, doff, n_reads, n_bases };
            kb.packed = hb.packed;
            kb.d_npos = dnp;
            kb.n_npos = hb.packed ? hb.n_npos : 0;
            kept_.push_back(kb);
            in_keep_call_ = true;
            try {
                if (hb.packed) map_device_packed(reinterpret_cast<const uint32_t*>(db), doff, n_reads, n_bases, dnp, kb.n_npos, nullptr, nullptr, stream_, true);
                else map_device_async(db, doff, n_reads, n_bases, nullptr, nullptr, stream_);
            } catch (...) {
                in_keep_call_ = false;
                throw;
            }
            in_keep_call_ = false;
            HIPCHK(hipEventSynchronize(kept_copied_)); // the caller's block is free again
            return;
        }
        // over the limit (or the device is full): nothing is kept from here on, and what was kept is of no use without the rest
        const uint64_t cap = kept_cap_;
        drop_kept();
        kept_cap_ = cap;
        kept_broken_ = true;
        // (said once per context, always: the run takes a different, slower route from here -- results are the same)
        std::fprintf(stderr, "[drprg-hip] the sample is larger than the %.1f GB of device memory set aside for resident reads (device %d): reads are "
                             "not kept, later passes read the file again (DRPRG_HIP_KEEP_READS_GB raises the limit)\n", (double)cap / 1e9, device_);
    }
    if (!use_filter_ || max_lanes_ > 1) { // no deferred form of this sequence
        map_host(hb);
        return;
    }
    HIPCHK(hipSetDevice(device_));
    if (offsets[0] != 0) throw Error(DRPRG_EINVAL, "offsets[0] must be 0");
    const uint64_t n_bases = hb.n_bases(), bytes = hb.payload_bytes();
    if (n_bases == 0) { // only empty reads: nothing to copy, nothing to map (the counters still see them)
        tot_reads_ += n_reads;
        return;
    }
    if (!copy_stream_) HIPCHK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    // the batch that used this staging set two calls ago was completed by the previous call (map_device_async completes the batch
    // before the one it queues), so the set is free
    Stage& st = stage_[stage_next_];
    stage_next_ ^= 1;
    if (!st.copied) HIPCHK(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
    if (bytes + 64 > st.bases_cap) {
        sync(); // (freeing device memory waits for the device; be explicit about the batch in flight)
        dfree(st.d_bases);
        st.bases_cap = bytes + bytes / 4 + 64;
        dmalloc(st.d_bases, st.bases_cap);
    }
    if (n_reads + 1 > st.reads_cap) {
        sync();
        dfree(st.d_offsets);
        st.reads_cap = n_reads + n_reads / 4 + 1;
        dmalloc(st.d_offsets, st.reads_cap);
    }
    if (hb.packed && hb.n_npos > st.npos_cap) {
        sync();
        dfree(st.d_npos);
        st.npos_cap = hb.n_npos + hb.n_npos / 4 + 16;
        dmalloc(st.d_npos, st.npos_cap);
    }
    HIPCHK(hipMemcpyAsync(st.d_bases, hb.bases, bytes, hipMemcpyHostToDevice, copy_stream_));
    HIPCHK(hipMemcpyAsync(st.d_offsets, offsets, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, copy_stream_));
    if (hb.packed && hb.n_npos) HIPCHK(hipMemcpyAsync(st.d_npos, hb.npos, hb.n_npos * sizeof(uint64_t), hipMemcpyHostToDevice, copy_stream_));
    HIPCHK(hipEventRecord(st.copied, copy_stream_));
    HIPCHK(hipStreamWaitEvent(stream_, st.copied, 0));
    if (hb.packed) map_device_packed(reinterpret_cast<const uint32_t*>(st.d_bases), st.d_offsets, n_reads, n_bases, hb.n_npos ? st.d_npos : nullptr, hb.n_npos, nullptr,
        nullptr, stream_, true);
    else map_device_async(st.d_bases, st.d_offsets, n_reads, n_bases, nullptr, nullptr, stream_);
    HIPCHK(hipEventSynchronize(st.copied)); // the caller's block is free again; the kernels run on
}

void Mapper::add_vectors_from(Mapper& other)
{
    if (other.n_knodes_ != n_knodes_ || other.n_prgs_ != n_prgs_) throw Error(DRPRG_EINVAL, "coverage size mismatch");
    other.sync();
    HIPCHK(hipSetDevice(other.device_));
    HIPCHK(hipStreamSynchronize(other.stream_));
    sync();
    HIPCHK(hipSetDevice(device_));
    const size_t nc = 2 * (size_t)n_knodes_, np = n_prgs_;
    const uint32_t* src = other.d_covg_; // [coverage | reads per PRG], one vector on either side
    if (other.device_ != device_) {
        if (!d_peer_tmp_) dmalloc(d_peer_tmp_, nc + np);
        HIPCHK(hipMemcpyPeerAsync(d_peer_tmp_, device_, other.d_covg_, other.device_, (nc + np) * sizeof(uint32_t), stream_));
        src = d_peer_tmp_;
    }
    HIPCHK(dev::launch_vector_add_u32(d_covg_, src, nc + np, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
}

void Mapper::download(std::vector<uint32_t>& covg, std::vector<uint32_t>& prg_reads)
{
    sync();
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipStreamSynchronize(stream_));
    covg.resize(2 * (size_t)n_knodes_);
    prg_reads.resize(n_prgs_);
    HIPCHK(hipMemcpy(covg.data(), d_covg_, covg.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(prg_reads.data(), d_prg_reads_, prg_reads.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
}

void Mapper::upload(const std::vector<uint32_t>& covg, const std::vector<uint32_t>& prg_reads)
{
    if (kept_cap_) kept_broken_ = true; // coverage that came without its reads

    if (covg.size() != 2 * (size_t)n_knodes_ || prg_reads.size() != n_prgs_) throw Error(DRPRG_EINVAL, "coverage size mismatch");
    sync();
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipMemcpy(d_covg_, covg.data(), covg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_prg_reads_, prg_reads.data(), prg_reads.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(nullptr));
}

void Mapper::device_tables(uint64_t out[6]) const
{
    out[4] = use_mid_ ? (uint64_t)MID_BITMAP_WORDS * 4 + ((uint64_t)16 << midc_wbits_) : 0; // global-memory (L2) tiers of the filter
    if (!use_mid_ && use_filter_ && d_blkc_) out[4] = (uint64_t)16 << blkc_wbits_; // (small tier: the second stage's block filter, what packed batches are tested against)
    out[5] = 0;
    out[0] = d_pbloom_ ? (uint64_t)1 << pbloom_wbits_ : 0;
    out[1] = ((uint64_t)1 << table_bits_) * (wide_hash_ ? 16 : 12);
    out[2] = bloom_wbits_ ? ((uint64_t)4 << bloom_wbits_) + (bloom0_wbits_ ? ((uint64_t)4 << bloom0_wbits_) + ((uint64_t)4 << BLOOMR_WBITS) : 0) : 0;
    if (use_mid_) out[2] = (uint64_t)4 << 15;
    out[3] = use_filter_ ? 2 : (use_direct_cands_ ? 3 : 1);
}

MapCounters Mapper::counters()
{
    sync();
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipStreamSynchronize(stream_));
    unsigned long long c[C_N];
    HIPCHK(hipMemcpy(c, d_counters_, sizeof(c), hipMemcpyDeviceToHost));
    if (std::getenv("DRPRG_WAVE_DEBUG"))
        std::fprintf(stderr, "[sketch_wave] entries %llu: several records out of group %llu, other group than the read's first %llu, read not inside the tile %llu, "
                             "more than 64 entries %llu; entries left to the records path %llu\n", c[C_CHUNK], c[C_CHUNK + 1], c[C_CHUNK + 2], c[C_CHUNK + 3], c[C_CHUNK + 4], c[C_CHUNK + 5]);
    if (d_ft_stat_) {
        unsigned long long st[4];
        HIPCHK(hipMemcpy(st, d_ft_stat_, sizeof(st), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[sketch_filter middle tier] groups %llu, past level 0 %llu (%.2f %%), past the bitmap %llu (%.2f %%), candidate positions %llu\n", st[0],
            st[1], st[0] ? 100.0 * (double)st[1] / (double)st[0] : 0.0, st[2], st[0] ? 100.0 * (double)st[2] / (double)st[0] : 0.0, st[3]);
    }
    MapCounters m;
    m.reads = tot_reads_;
    m.bases = tot_bases_;
    m.minimizers = tot_minimizers_;
    m.hits = tot_hits_;
    m.clusters_kept = c[C_CLUSTERS_KEPT];
    m.hits_kept = c[C_HITS_KEPT];
    m.kernel = use_filter_ ? 2 : (use_direct_cands_ ? 3 : 1);
    m.leftover_reads = tot_leftover_;
    return m;
}

} // namespace drprg
