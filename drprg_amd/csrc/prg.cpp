// prg.cpp -- PRG string parser and local-graph helpers.  See prg.h.
#include "prg.h"
#include <cctype>
#include <fstream>
#include <functional>
#include <set>

namespace drprg {

namespace {
struct Tok {
    bool marker;
    int value;          // marker number
    uint32_t start, end; // for sequence segments: [start,end) in the PRG string
};

// Split the PRG string into an alternation seg, marker, seg, marker, ..., seg (segments may be empty).
std::vector<Tok> tokenize(const std::string& s)
{
    std::vector<Tok> toks;
    const uint32_t n = (uint32_t)s.size();
    uint32_t seg_start = 0, i = 0;
    auto push_seg = [&](uint32_t a, uint32_t b) {
        if (b < a) b = a;
        if (a > n) a = b = n;
        toks.push_back(Tok { false, 0, a, b });
    };
    while (i < n) {
        if (std::isdigit((unsigned char)s[i])) {
            uint32_t ds = i;
            int v = 0;
            while (i < n && std::isdigit((unsigned char)s[i])) {
                v = v * 10 + (s[i] - '0');
                ++i;
            }
            // the marker owns one space on each side
            uint32_t seg_end = ds > 0 && s[ds - 1] == ' ' ? ds - 1 : ds;
            push_seg(seg_start, seg_end);
            toks.push_back(Tok { true, v, ds, i });
            seg_start = (i < n && s[i] == ' ') ? i + 1 : i;
            if (i < n && s[i] == ' ') ++i;
        } else {
            ++i;
        }
    }
    push_seg(seg_start, n);
    return toks;
}
} // namespace

void LocalGraph::parse(const std::string& nm, const std::string& prg_string)
{
    name = nm;
    prg = prg_string;
    nodes.clear();
    chains.clear();
    sites.clear();
    for (char c : prg)
        if (!(std::isdigit((unsigned char)c) || c == ' ' || nt4((unsigned char)c) < 4))
            throw Error(DRPRG_EFORMAT, "PRG '" + name + "' holds a character that is not ACGT, digit or space");
    const std::vector<Tok> toks = tokenize(prg);

    auto add_node = [&](const Tok& t, const std::vector<uint32_t>& from, int chain) {
        LocalNode nd;
        nd.id = (uint32_t)nodes.size();
        nd.start = t.start;
        nd.end = t.end;
        nd.seq = prg.substr(t.start, t.end - t.start);
        for (char& c : nd.seq) c = (char)std::toupper((unsigned char)c);
        if (nd.seq.find(' ') != std::string::npos)
            throw Error(DRPRG_EFORMAT, "PRG '" + name + "': unexpected space inside a sequence segment");
        nd.chain = chain;
        nodes.push_back(nd);
        for (uint32_t f : from) {
            nodes[f].out.push_back(nd.id);
            nodes[nd.id].in.push_back(f);
        }
        return nd.id;
    };

    // Recursive descent.  parse_chain consumes tokens starting at a segment token and stops at the
    // marker `stop_odd` / `stop_even` (or at the end for the top level).
    std::function<std::vector<uint32_t>(size_t&, std::vector<uint32_t>, int, int, int, int)> parse_chain;
    parse_chain = [&](size_t& idx, std::vector<uint32_t> from, int stop_odd, int stop_even, int parent_site,
                      int level) -> std::vector<uint32_t> {
        int chain_id = (int)chains.size();
        chains.emplace_back();
        chains[chain_id].parent_site = parent_site;
        while (true) {
            if (idx >= toks.size() || toks[idx].marker)
                throw Error(DRPRG_EFORMAT, "PRG '" + name + "': malformed marker structure");
            uint32_t nid = add_node(toks[idx], from, chain_id);
            chains[chain_id].nodes.push_back(nid);
            ++idx;
            if (idx >= toks.size()) {
                if (stop_odd >= 0)
                    throw Error(DRPRG_EFORMAT, "PRG '" + name + "': site " + std::to_string(stop_odd) + " is not closed");
                return { nid };
            }
            int m = toks[idx].value;
            if (m == stop_odd || m == stop_even) return { nid };
            if (m % 2 == 0)
                throw Error(DRPRG_EFORMAT, "PRG '" + name + "': allele separator " + std::to_string(m) + " outside its site");
            // a new site opens
            int site_id = (int)sites.size();
            sites.emplace_back();
            sites[site_id].marker = m;
            sites[site_id].level = level;
            sites[site_id].parent_chain = chain_id;
            sites[site_id].pre_node = nid;
            chains[chain_id].sites.push_back(site_id);
            ++idx;
            std::vector<uint32_t> ends;
            while (true) {
                int allele_chain = (int)chains.size();
                std::vector<uint32_t> e = parse_chain(idx, { nid }, m, m + 1, site_id, level + 1);
                sites[site_id].alleles.push_back(allele_chain);
                ends.insert(ends.end(), e.begin(), e.end());
                int stop = toks[idx].value;
                ++idx;
                if (stop == m) break;
            }
            if (sites[site_id].alleles.size() < 2)
                throw Error(DRPRG_EFORMAT, "PRG '" + name + "': site " + std::to_string(m) + " has a single allele");
            from = ends;
            // next loop iteration creates the post-site node
            if (idx < toks.size() && !toks[idx].marker) sites[site_id].post_node = (uint32_t)nodes.size();
        }
    };
    size_t idx = 0;
    parse_chain(idx, {}, -1, -1, -1, 0);
    if (idx < toks.size())
        throw Error(DRPRG_EFORMAT, "PRG '" + name + "': trailing tokens after the top-level chain");
}

std::vector<uint32_t> LocalGraph::top_path() const
{
    std::vector<uint32_t> p;
    std::function<void(int)> walk = [&](int c) {
        const Chain& ch = chains[c];
        for (size_t i = 0; i < ch.nodes.size(); ++i) {
            p.push_back(ch.nodes[i]);
            if (i < ch.sites.size()) walk(sites[ch.sites[i]].alleles[0]);
        }
    };
    walk(0);
    return p;
}

std::string LocalGraph::string_along_path(const std::vector<uint32_t>& path) const
{
    std::string s;
    for (uint32_t n : path) s += nodes[n].seq;
    return s;
}

std::vector<uint32_t> LocalGraph::nodes_along_string(const std::string& query) const
{
    std::string q = query;
    for (char& c : q) c = (char)std::toupper((unsigned char)c);
    std::set<std::pair<uint32_t, uint32_t>> dead; // (node, offset) known not to reach the sink
    std::vector<uint32_t> path;
    const uint32_t snk = sink();
    std::function<bool(uint32_t, uint32_t)> dfs = [&](uint32_t n, uint32_t off) -> bool {
        if (dead.count({ n, off })) return false;
        const LocalNode& nd = nodes[n];
        if (off + nd.len() > q.size() || q.compare(off, nd.len(), nd.seq) != 0) {
            dead.insert({ n, off });
            return false;
        }
        path.push_back(n);
        uint32_t noff = off + nd.len();
        if (n == snk || nd.out.empty()) {
            if (noff == q.size()) return true;
        } else {
            for (uint32_t o : nd.out)
                if (dfs(o, noff)) return true;
        }
        path.pop_back();
        dead.insert({ n, off });
        return false;
    };
    if (nodes.empty() || !dfs(0, 0)) return {};
    return path;
}

std::vector<LocalGraph> load_prg_file(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw Error(DRPRG_ENOENT, "cannot open PRG file " + path);
    std::vector<LocalGraph> out;
    std::string line, name, seq;
    bool have = false;
    auto flush = [&]() {
        if (!have) return;
        LocalGraph g;
        g.parse(name, seq);
        out.push_back(std::move(g));
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            flush();
            have = true;
            size_t e = line.find_first_of(" \t");
            name = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
            seq.clear();
        } else {
            seq += line;
        }
    }
    flush();
    if (out.empty()) throw Error(DRPRG_EFORMAT, "no PRG records in " + path);
    return out;
}

} // namespace drprg
