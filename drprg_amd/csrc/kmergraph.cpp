// kmergraph.cpp -- minimizer sketch of a PRG local graph.  See kmergraph.h and DESIGN.md "Semantics".
//
// Definition used (equivalent on every linear walk to the read sketch of kernels.hip / oracle.c):
// a k-mer path is a k-mer-graph node iff it is a window minimizer of some walk through the local
// graph.  Construction is forward-greedy, as in pandora: from a minimizer m, the next minimizer
// on a walk is the first later k-mer (within w-1 shifts) whose hash is <= hash(m); if there is
// none, it is the leftmost minimum of the w k-mers that follow m.
#include "kmergraph.h"
#include <algorithm>
#include <deque>
#include <fstream>
#include <sstream>

namespace drprg {

namespace {

struct Ext {
    std::vector<PathPiece> empties; // empty local nodes crossed before reaching the base
    uint32_t node, off;
};

// `at_end` is set when some route from `node` reaches the end of the graph without reading another base (the node itself
// is the last one, or only empty nodes follow on that route)
void extensions_rec(const LocalGraph& g, uint32_t node, std::vector<PathPiece>& prefix, std::vector<Ext>& out, bool& at_end)
{
    if (g.nodes[node].out.empty()) at_end = true;
    for (uint32_t o : g.nodes[node].out) {
        if (g.nodes[o].len() > 0) {
            out.push_back(Ext { prefix, o, 0 });
        } else {
            prefix.push_back(PathPiece { o, 0, 0 });
            extensions_rec(g, o, prefix, out, at_end);
            prefix.pop_back();
        }
    }
}

// all ways to read one more base after the last base of `p`; returns true if a walk may also end here
bool extensions(const LocalGraph& g, const KPath& p, std::vector<Ext>& out)
{
    out.clear();
    const PathPiece& last = p.back();
    if (last.off_end < g.nodes[last.node].len()) {
        out.push_back(Ext { {}, last.node, last.off_end });
        return false;
    }
    std::vector<PathPiece> prefix;
    bool at_end = false;
    extensions_rec(g, last.node, prefix, out, at_end);
    return at_end;
}

void append_base(KPath& p, const Ext& e)
{
    if (!p.empty() && e.empties.empty() && p.back().node == e.node && p.back().off_end == e.off) {
        p.back().off_end++;
        return;
    }
    for (const PathPiece& pp : e.empties) p.push_back(pp);
    p.push_back(PathPiece { e.node, e.off, e.off + 1 });
}

void drop_first_base(KPath& p)
{
    p.front().off_start++;
    size_t i = 0;
    while (i < p.size() && p[i].off_start == p[i].off_end) ++i;
    p.erase(p.begin(), p.begin() + (long)i);
}

struct Cand {
    KPath path;
    uint64_t hash;
    bool strand;
};

struct Builder {
    const LocalGraph& g;
    KmerGraph& kg;
    int w, k;
    std::map<KPath, uint32_t> ids;
    std::deque<uint32_t> todo;

    uint32_t get_node(const Cand& c)
    {
        auto it = ids.find(c.path);
        if (it != ids.end()) return it->second;
        KmerNode n;
        n.id = (uint32_t)kg.nodes.size();
        n.path = c.path;
        n.hash = c.hash;
        n.strand = c.strand;
        kg.nodes.push_back(n);
        ids[c.path] = n.id;
        todo.push_back(n.id);
        return n.id;
    }
    void link(uint32_t a, uint32_t b)
    {
        auto& o = kg.nodes[a].out;
        if (std::find(o.begin(), o.end(), b) == o.end()) {
            o.push_back(b);
            kg.nodes[b].in.push_back(a);
        }
    }
    Cand make_cand(const KPath& p)
    {
        Cand c;
        c.path = p;
        std::string s = kpath_sequence(g, p);
        canonical_kmer_hash(s.data(), k, c.hash, c.strand);
        return c;
    }
    void link_leftmost_min(uint32_t from, const std::vector<Cand>& cands)
    {
        size_t best = 0;
        for (size_t i = 1; i < cands.size(); ++i)
            if (cands[i].hash < cands[best].hash) best = i;
        link(from, get_node(cands[best]));
    }

    // slide one base at a time from `cur`; `from` is the minimizer being expanded (0 = source)
    void slide(uint32_t from, const KPath& cur, std::vector<Cand>& cands)
    {
        std::vector<Ext> exts;
        if (extensions(g, cur, exts)) { // a walk ends here (the last node, or through empty alleles up to the end of the PRG)
            if (from == 0) {
                if (!cands.empty()) link_leftmost_min(0, cands);
            } else {
                link(from, SINK);
            }
        }
        for (const Ext& e : exts) {
            KPath np = cur;
            append_base(np, e);
            drop_first_base(np);
            Cand c = make_cand(np);
            if (from != 0 && c.hash <= kg.nodes[from].hash) {
                link(from, get_node(c));
                continue;
            }
            cands.push_back(c);
            if ((int)cands.size() == w) {
                link_leftmost_min(from, cands);
            } else {
                slide(from, np, cands);
            }
            cands.pop_back();
        }
    }

    // collect the first k bases along every walk from the graph start
    void first_kmers(KPath& p, uint32_t have)
    {
        if ((int)have == k) {
            std::vector<Cand> cands { make_cand(p) };
            if (w == 1) link_leftmost_min(0, cands);
            else slide(0, p, cands);
            return;
        }
        std::vector<Ext> exts;
        if (p.empty()) {
            if (g.nodes[0].len() > 0) exts.push_back(Ext { {}, 0, 0 });
            else {
                std::vector<PathPiece> prefix;
                bool at_end = false;
                extensions_rec(g, 0, prefix, exts, at_end);
                for (Ext& e : exts) e.empties.clear(); // a k-mer path never starts with empty pieces
            }
        } else {
            extensions(g, p, exts);
        }
        for (const Ext& e : exts) {
            KPath np = p;
            append_base(np, e);
            first_kmers(np, have + 1);
        }
    }

    static constexpr uint32_t SINK = 1; // temporary id during construction
};
} // namespace

std::string kpath_sequence(const LocalGraph& g, const KPath& p)
{
    std::string s;
    for (const PathPiece& pp : p) s.append(g.nodes[pp.node].seq, pp.off_start, pp.off_end - pp.off_start);
    return s;
}

uint32_t kpath_start_coord(const LocalGraph& g, const KPath& p)
{
    return p.empty() ? 0 : g.nodes[p.front().node].start + p.front().off_start;
}

void KmerGraph::build(const LocalGraph& g, int w_, int k_)
{
    w = w_;
    k = k_;
    nodes.clear();
    nodes.resize(2); // 0 = source, 1 = sink (renumbered last by finalize)
    nodes[0].id = 0;
    nodes[1].id = 1;
    Builder b { g, *this, w, k, {}, {} };
    KPath p;
    b.first_kmers(p, 0);
    while (!b.todo.empty()) {
        uint32_t id = b.todo.front();
        b.todo.pop_front();
        std::vector<Cand> cands;
        KPath cur = nodes[id].path;
        b.slide(id, cur, cands);
    }
    if (nodes[0].out.empty()) { // PRG shorter than k: source -> sink
        nodes[0].out.push_back(1);
        nodes[1].in.push_back(0);
    }
    // renumber: source, k-mers by (start coordinate, end coordinate, path), sink
    struct Key {
        uint32_t start, end;
        const KPath* path;
        uint32_t old;
    };
    std::vector<Key> keys;
    for (uint32_t i = 2; i < nodes.size(); ++i) {
        const KPath& kp = nodes[i].path;
        keys.push_back(Key { kpath_start_coord(g, kp), g.nodes[kp.back().node].start + kp.back().off_end, &kp, i });
    }
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        if (a.start != b.start) return a.start < b.start;
        if (a.end != b.end) return a.end < b.end;
        return *a.path < *b.path;
    });
    std::vector<uint32_t> remap(nodes.size());
    remap[0] = 0;
    remap[1] = (uint32_t)nodes.size() - 1;
    for (uint32_t i = 0; i < keys.size(); ++i) remap[keys[i].old] = i + 1;
    std::vector<KmerNode> sorted(nodes.size());
    for (uint32_t i = 0; i < nodes.size(); ++i) {
        KmerNode n = std::move(nodes[i]);
        n.id = remap[i];
        for (uint32_t& o : n.out) o = remap[o];
        for (uint32_t& o : n.in) o = remap[o];
        std::sort(n.out.begin(), n.out.end());
        std::sort(n.in.begin(), n.in.end());
        sorted[n.id] = std::move(n);
    }
    nodes = std::move(sorted);
    finalize();
}

void KmerGraph::finalize()
{
    const uint32_t n = (uint32_t)nodes.size();
    for (const KmerNode& nd : nodes)
        for (uint32_t o : nd.out)
            if (o <= nd.id) throw Error(DRPRG_EFORMAT, "k-mer graph node ids are not a topological order");
    std::vector<uint32_t> len(n, 0);
    for (uint32_t j = n - 1; j-- > 0;) {
        uint32_t best = 0;
        bool any = false;
        for (uint32_t o : nodes[j].out) {
            uint32_t cand = len[o] + 1;
            if (!any || cand < best) { best = cand; any = true; }
        }
        len[j] = any ? best : 0;
    }
    shortest_path_length = n ? len[0] : 0;
}

static std::string path_to_string(const LocalGraph& g, const KPath& p)
{
    std::ostringstream os;
    os << p.size() << "{";
    for (const PathPiece& pp : p) {
        uint32_t base = g.nodes[pp.node].start;
        os << "[" << base + pp.off_start << ", " << base + pp.off_end << ")";
    }
    os << "}";
    return os.str();
}

void KmerGraph::save_gfa(const std::string& path, const LocalGraph& g) const
{
    std::ofstream out(path);
    if (!out) throw Error(DRPRG_EIO, "cannot write " + path);
    out << "H\tVN:Z:1.0\tbn:Z:--linear --singlearr\n";
    for (const KmerNode& n : nodes) {
        out << "S\t" << n.id << "\t" << path_to_string(g, n.path) << "\tFC:i:0\tRC:i:0\n";
        for (uint32_t o : n.out) out << "L\t" << n.id << "\t+\t" << o << "\t+\t0M\n";
    }
    if (!out) throw Error(DRPRG_EIO, "short write to " + path);
}

void KmerGraph::load_gfa(const std::string& path, const LocalGraph& g, int w_, int k_)
{
    std::ifstream in(path);
    if (!in) throw Error(DRPRG_ENOENT, "cannot open " + path);
    w = w_;
    k = k_;
    nodes.clear();
    // coordinate -> local node lookups
    std::map<uint32_t, uint32_t> empty_at, start_of; // start coordinate -> node id
    for (const LocalNode& n : g.nodes) {
        if (n.len() == 0) empty_at[n.start] = n.id;
        else start_of[n.start] = n.id;
    }
    std::string line;
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        if (line[0] == 'S') {
            std::istringstream is(line);
            std::string tag, pstr;
            uint32_t id;
            is >> tag >> id;
            std::getline(is, pstr, '\t'); // empty (between id and path)
            std::getline(is, pstr, '\t');
            if (id != nodes.size()) throw Error(DRPRG_EFORMAT, path + ": S lines out of order");
            KmerNode n;
            n.id = id;
            size_t pos = pstr.find('{');
            if (pos == std::string::npos) throw Error(DRPRG_EFORMAT, path + ": malformed path " + pstr);
            while ((pos = pstr.find('[', pos)) != std::string::npos) {
                unsigned a = 0, b = 0;
                if (std::sscanf(pstr.c_str() + pos, "[%u, %u)", &a, &b) != 2)
                    throw Error(DRPRG_EFORMAT, path + ": malformed interval in " + pstr);
                ++pos;
                if (a == b) {
                    auto it = empty_at.find(a);
                    if (it == empty_at.end()) throw Error(DRPRG_EFORMAT, path + ": empty interval matches no PRG node");
                    n.path.push_back(PathPiece { it->second, 0, 0 });
                } else {
                    auto it = start_of.upper_bound(a);
                    if (it == start_of.begin()) throw Error(DRPRG_EFORMAT, path + ": interval matches no PRG node");
                    --it;
                    const LocalNode& ln = g.nodes[it->second];
                    if (b > ln.end) throw Error(DRPRG_EFORMAT, path + ": interval crosses a PRG node boundary");
                    n.path.push_back(PathPiece { ln.id, a - ln.start, b - ln.start });
                }
            }
            if (!n.path.empty()) {
                std::string s = kpath_sequence(g, n.path);
                if ((int)s.size() != k || !canonical_kmer_hash(s.data(), k, n.hash, n.strand))
                    throw Error(DRPRG_EFORMAT, path + ": node " + std::to_string(id) + " is not a " + std::to_string(k) + "-mer");
            }
            nodes.push_back(std::move(n));
        } else if (line[0] == 'L') {
            unsigned a, b;
            char s1, s2;
            if (std::sscanf(line.c_str(), "L\t%u\t%c\t%u\t%c", &a, &s1, &b, &s2) != 4)
                throw Error(DRPRG_EFORMAT, path + ": malformed L line");
            edges.push_back({ a, b });
        }
    }
    for (auto& e : edges) {
        if (e.first >= nodes.size() || e.second >= nodes.size()) throw Error(DRPRG_EFORMAT, path + ": edge to unknown node");
        nodes[e.first].out.push_back(e.second);
        nodes[e.second].in.push_back(e.first);
    }
    if (nodes.size() < 2) throw Error(DRPRG_EFORMAT, path + ": k-mer graph has no source/sink");
    finalize();
}

} // namespace drprg
