// index.h -- minimizer index of a set of PRGs: hash -> (prg, k-mer node, strand) records, plus the
// flattened tables the HIP kernels probe.
//
// Stands in for pandora's Index (`pandora index`, reference call site
// /root/reference/src/lib.rs:479-510; on-disk names /root/reference/src/builder.rs:263-269:
// `<prg>.k<K>.w<W>.idx` and `kmer_prgs/`).
#pragma once
#include "kmergraph.h"

namespace drprg {

struct MiniRecord {
    uint32_t prg;
    uint32_t knode; // k-mer node id local to the PRG's k-mer graph
    uint8_t strand; // 1 = the node's forward k-mer is the canonical one
};

// Flat, device-friendly view of the index (all arrays owned by PrgIndex).
struct FlatIndex {
    // sorted distinct keys + CSR records (what the oracle consumes)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> rec_off; // keys.size()+1
    std::vector<uint32_t> rec_prg, rec_knode_global;
    std::vector<uint8_t> rec_strand;
    // per-PRG
    std::vector<uint32_t> knode_base;    // prefix sum of k-mer-graph sizes (nprg+1)
    std::vector<uint32_t> min_path_len;  // KmerGraph::shortest_path_length
    // open-addressed probe table: slot = {key, rec_off, rec_cnt}; empty slot has cnt == 0
    uint32_t table_bits = 0;
    std::vector<uint64_t> slot_key;
    std::vector<uint32_t> slot_off, slot_cnt;
    // Bloom filter over the 2-bit codes of every index k-mer in both orientations (k <= 15 only): two levels, each
    // 3 bits in one 32-bit word of the same array (layout at the top of sketch_filter_kernel, sketch_filter.hip).
    // bloom_wbits == 0: no filter (k > 15, or too many keys for an LDS-resident filter).
    uint32_t bloom_wbits = 0;
    std::vector<uint32_t> bloom;
    // level 0 of that filter (k = 15, small indexes): the 12-mers at offsets 0..3 of every index k-mer; 0 = absent
    uint32_t bloom0_wbits = 0;
    std::vector<uint32_t> bloom0;
    std::vector<uint32_t> bloom0f; // bloom0 plus the second-stage bits of every code (four per code, in the word its BLOOM_CR hash selects)
    // with level 0: the second stage (refine_kernel), 2^14 words keyed on the whole k-mer code, four bits per code
    std::vector<uint32_t> bloomr;
    // middle tier (k = 15, too many index k-mers for the forms above; common.h MID_*): level 0 keyed on CANONICAL 12-mers (its own
    // array `mid0`, 2^15 words, `mid0_bits` = 3 or 1 bits per 12-mer), the exact bitmap of those canonical 12-mers, and the
    // split-block Bloom filter of the whole codes (2^midc_wbits blocks of four words, block chosen by the 12-mer).  midc_wbits == 0: absent.
    uint32_t midc_wbits = 0, mid0_bits = 0;
    std::vector<uint32_t> mid0, mid_bitmap, midc;
    // Small tier (level 0 present), round 6: the second stage as a split-block filter of the codes for global memory (2^blkc_wbits blocks of four
    // words, <= 4 entries a block; block = hash of the PLAIN 12-mer at offset o of the code, o = 0..3; a code sets one bit in each word of each of
    // its four blocks) -- what sketch_filter_kernel<.., MID = 2> tests a group against, so that `bloom0` can stay level 0 alone.  0: absent.
    uint32_t blkc_wbits = 0;
    std::vector<uint32_t> blkc;
    uint32_t total_knodes() const { return knode_base.empty() ? 0 : knode_base.back(); }
};

struct PrgIndex {
    int w = 0, k = 0;
    std::vector<LocalGraph> prgs;
    std::vector<KmerGraph> kgs;
    FlatIndex flat;
    // Host restatement of the kernels' filter tests on every index k-mer (both orientations).  out[0] = codes tested,
    // out[1..4] = codes that level 0 / levels 1+2 / the second stage of the level-0 form / nothing at all would let
    // through... i.e. out[1..3] count FALSE NEGATIVES (must be 0); out[4..6] = bits set per thousand in bloom0 / bloom / bloomr.
    void filter_selfcheck(uint64_t out[8]) const;

    // `pandora index`: sketch every PRG of prg_file, write <prg_file>.k<k>.w<w>.idx and
    // <dir>/kmer_prgs/<name>.k<k>.w<w>.gfa
    static void build_and_save(const std::string& prg_file, int w, int k, int threads);
    // in-memory build (no files touched)
    void build(const std::string& prg_file, int w, int k, int threads);
    // load what build_and_save wrote
    void load(const std::string& prg_file, int w, int k);
    void save(const std::string& prg_file) const;

    static std::string idx_path(const std::string& prg_file, int w, int k);
    static std::string gfa_path(const std::string& prg_file, const std::string& name, int w, int k);

private:
    void flatten();
};

// slot index of `key` in a table of 2^bits slots (multiplicative hash; same function on the device).  narrow: the keys are
// 32-bit hashes (k <= 15) and the device does the whole thing in one 32-bit multiply -- the same product that picks the word
// and the bits of the Bloom tier in front of a large table (kernels.h pbloom_mix)
inline uint32_t table_slot(uint64_t key, uint32_t bits, bool narrow)
{
    if (narrow) return ((uint32_t)key * 0x9E3779B1u) >> (32 - bits);
    return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - bits));
}

} // namespace drprg
