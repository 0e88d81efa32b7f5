// prg.h -- PRG string -> local graph (one per locus of dr.prg).
//
// Restates pandora's LocalPRG::build_graph (external; the reference only ships the file format,
// /root/reference/tests/cases/expected/dr.prg, and the node-interval convention visible in
// /root/reference/src/lib.rs:3009-3050: intervals are character offsets into the PRG string
// *including* marker digits and their spaces).
#pragma once
#include "common.h"
#include <map>

namespace drprg {

struct LocalNode {
    uint32_t id = 0;
    uint32_t start = 0, end = 0; // [start,end) offsets into the PRG string
    std::string seq;
    std::vector<uint32_t> out, in;
    int chain = -1; // Chain (allele / top level) this node belongs to
    uint32_t len() const { return end - start; }
};

// A Chain is an alternation node, site, node, site, ..., node: either the top level of the PRG
// or one allele of a site.
struct Chain {
    std::vector<uint32_t> nodes;
    std::vector<int> sites; // sites.size() == nodes.size() - 1
    int parent_site = -1;   // -1 for top level
};

struct Site {
    int marker = 0; // odd number opening/closing the site
    int level = 0;  // 0 = top level
    int parent_chain = 0;
    uint32_t pre_node = 0, post_node = 0;
    std::vector<int> alleles; // chain ids
};

struct LocalGraph {
    std::string name;
    std::string prg; // the raw PRG string
    std::vector<LocalNode> nodes;
    std::vector<Chain> chains; // chains[0] = top level
    std::vector<Site> sites;

    void parse(const std::string& name, const std::string& prg_string);
    uint32_t sink() const { return (uint32_t)nodes.size() - 1; }
    // Path (node ids, empty nodes included) from node 0 to the sink that spells `s` exactly;
    // empty vector if none (pandora LocalPRG::nodes_along_string).
    std::vector<uint32_t> nodes_along_string(const std::string& s) const;
    // The path that takes the first allele at every site.
    std::vector<uint32_t> top_path() const;
    std::string string_along_path(const std::vector<uint32_t>& path) const;
};

// dr.prg: FASTA, one PRG string per record, sequence on one line (multi-line tolerated).
std::vector<LocalGraph> load_prg_file(const std::string& path);

} // namespace drprg
