// kmergraph.h -- (w,k)-minimizer sketch of a local PRG graph: the k-mer graph whose nodes carry
// the per-sample forward/reverse coverage that the GPU accumulates.
//
// Restates pandora's LocalPRG::minimizer_sketch / KmerGraph (external program run by
// `pandora index`, reference call site /root/reference/src/lib.rs:479-510; outputs named at
// /root/reference/src/builder.rs:263-269).
#pragma once
#include "prg.h"

namespace drprg {

// One piece of a k-mer's walk through the local graph: bases [off_start, off_end) of `node`.
// Empty local nodes crossed by the walk are kept as zero-length pieces.
struct PathPiece {
    uint32_t node, off_start, off_end;
    bool operator<(const PathPiece& o) const
    {
        if (node != o.node) return node < o.node;
        if (off_start != o.off_start) return off_start < o.off_start;
        return off_end < o.off_end;
    }
    bool operator==(const PathPiece& o) const { return node == o.node && off_start == o.off_start && off_end == o.off_end; }
};
using KPath = std::vector<PathPiece>;

struct KmerNode {
    uint32_t id = 0;
    KPath path;        // empty for source (id 0) and sink (last id)
    uint64_t hash = 0; // canonical minimizer hash
    bool strand = true; // forward k-mer is the canonical one
    std::vector<uint32_t> out, in;
};

struct KmerGraph {
    int w = 0, k = 0;
    std::vector<KmerNode> nodes; // id order is a topological order; 0 = source, last = sink
    uint32_t shortest_path_length = 0; // edges on the shortest source->sink path (pandora min_path_length)

    void build(const LocalGraph& g, int w, int k);
    void finalize(); // sort/renumber, compute shortest_path_length
    uint32_t n_kmers() const { return nodes.size() < 2 ? 0 : (uint32_t)nodes.size() - 2; }

    // kmer_prgs/<name>.k<k>.w<w>.gfa
    void save_gfa(const std::string& path, const LocalGraph& g) const;
    void load_gfa(const std::string& path, const LocalGraph& g, int w, int k);
};

std::string kpath_sequence(const LocalGraph& g, const KPath& p);
// PRG-string coordinate of the first base of a k-mer path
uint32_t kpath_start_coord(const LocalGraph& g, const KPath& p);

} // namespace drprg
