// filter_common.h -- constants and helpers shared by the kernels of the filtered launch sequence
// (sketch_filter.hip -> candidates.hip -> read_cluster.hip).
#pragma once
#include "common.h"
#include "device_common.h"

namespace drprg {
namespace dev {

constexpr int FT_THREADS = 1024;
constexpr int FT_WAVES = FT_THREADS / 64;
constexpr int FT_G = 32;                // positions per lane (ASCII input, and the forms without level 0)
constexpr int FT_WPOS = 63 * FT_G;      // positions per wave tile: lane 63's word is only lane 62's right neighbour
// the level-0 form on packed input takes 64 positions per lane: one 16-byte load per lane and tile
DRPRG_HD constexpr int filter_positions_per_lane(bool level0, bool packed) { return level0 && packed ? 64 : 32; }
constexpr int FT_BLOOM_WORDS = 1 << 14; // levels 1+2 of the filter, at most (64 KB of LDS)
constexpr int FT_L0_WORDS = 1 << 15;    // level 0 (128 KB; levels 1+2 then get 32 KB: all 160 KB of a CU)
constexpr int EX_THREADS = 256;
constexpr int SCAN_THREADS = 1024;
// Candidate slices (round 6): a slice holds the candidates of one CHUNK of consecutive wave tiles; sketch_filter_kernel's waves take their
// first chunk by number and the later ones from a device counter (FilterSched, kernels.h), so a wave fills as many slices as it takes chunks
// and the concatenation of the slices in chunk order is still the candidate list in position order.  FT_SUPER consecutive slices form a
// superblock whose (clamped) candidate count the filter kernel keeps as well: verify_scan_kernel scans the superblocks -- MAX_SLICES of
// them at most, as it scanned the slices themselves until round 5 -- and a candidate then finds its slice among the FT_SUPER of its superblock.
constexpr int MAX_SLICES = SCAN_THREADS * 8; // superblocks
#ifndef DRPRG_FT_DEPTH
#define DRPRG_FT_DEPTH 3
#endif
// tiles a wave of sketch_filter_kernel has in flight behind the one it works on.  3 since round 6 (a fourth register set: the kernel's budget is
// 128 VGPRs whatever it does -- 16 waves per CU -- and it uses 66-79): 4 kb reads 1.48-1.56 -> 1.44-1.45 ms per launch, 150-base reads and packed
// input unchanged (profiles/r06/schedule.txt 7).  Round 2 had found a fourth set slower -- with the group records still going to global memory.
constexpr int FT_DEPTH = DRPRG_FT_DEPTH;
constexpr int FT_SUPER = 8;                  // slices per superblock
constexpr int MAX_CHUNKS = MAX_SLICES * FT_SUPER; // slices
constexpr int MAX_EX_WG = SCAN_THREADS * 4;  // workgroups of verify_count_kernel / expand_kernel

// the contiguous range of the ordered candidate list that workgroup `wg` of `n_wg` owns
__device__ inline void candidate_range(const FilterWork& fw, uint32_t wg, uint32_t n_wg, uint32_t& t_begin, uint32_t& t_end)
{
    const uint32_t total = *fw.cand_total;
    const uint32_t per_wg = (total + n_wg - 1) / n_wg;
    const uint64_t b = (uint64_t)wg * per_wg;
    t_begin = b < total ? (uint32_t)b : total;
    t_end = b + per_wg < total ? (uint32_t)(b + per_wg) : total;
}

// stages of the filtered sequence that live in the other translation units (all asynchronous on `stream`)
// candidates.hip: slices -> dense ordered candidate list -> verified candidates + per-candidate records + batch totals
// (fw.verify_grid receives the number of workgroups whose totals the stage left in fw.wg_hits / wg_nmin / wg_maxlen)
hipError_t launch_candidate_stage(const SketchArgs& a, FilterWork& fw, const ReadClusterArgs& rc, int n_cus, hipStream_t stream, bool with_totals = true);
// read_verify.hip: the verification of a short-read batch read by read (k = 15, w in {11, 14}): every read that holds several candidates is
// sketched ONCE by a wave (sketch_block.h) instead of once per candidate; same outputs as verify_count_kernel
bool read_verify_applies(const SketchArgs& a, const FilterWork& fw);
uint32_t read_verify_grid(int n_cus);
hipError_t launch_read_verify(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc, uint32_t grid, hipStream_t stream);
// read_cluster.hip: per-read clustering straight from the candidate list (skip: mark the batch as left over instead)
hipError_t launch_read_cluster(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc, int n_cus, bool skip, hipStream_t stream);
// read_cluster_wave.hip: the same for the reads that fit a wave's 128 staged candidates (one wave per 64 candidates, no workgroup barrier)
hipError_t launch_read_cluster_wave(const SketchArgs& a, const FilterWork& fw, const ReadClusterArgs& rc, int n_cus, hipStream_t stream);

} // namespace dev
} // namespace drprg
