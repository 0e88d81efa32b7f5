/*
 * oracle_index.c -- CPU restatement of `pandora index` for one PRG (TEST INFRASTRUCTURE ONLY, part of liboracle.so).
 *
 * Row a-4 of SURVEY.md section 8: Pandora::index_with, /root/reference/src/lib.rs:479-510, spawns
 * `pandora index -t T -w W -k K <prg>`, which turns every PRG string of dr.prg into a k-mer graph (the nodes whose
 * coverage `pandora map` accumulates) and a minimizer -> (prg, k-mer node, strand) table; outputs are named at
 * /root/reference/src/builder.rs:263-269.  pandora's source is absent from /root/reference, so -- like oracle.c --
 * this is a restatement of its published algorithm (LocalPRG::build_graph, LocalPRG::minimizer_sketch,
 * KmerGraph::min_path_length).  Parity with pandora: the NODE SET is pinned since round 4 by the per-allele k-mer counts the
 * reference's fixture VCFs leak (tests/test_kmer_count_kat.py: the forward-greedy nodes included); node numbering, edges and
 * min_path_length stay unpinned.  What else the reference pins, and what the tests check this file against: the PRG string syntax (/root/reference/tests/cases/expected/dr.prg) and the node
 * interval convention (character offsets into the PRG string, markers and their spaces included:
 * /root/reference/src/lib.rs:3009-3050).
 *
 * It shares no code with the product's builder (drprg_amd/csrc/prg.cpp, kmergraph.cpp, index.cpp) and is written the
 * naive way round: the product slides a k-mer base by base through the graph with recursion and pruning; here every
 * walk fragment is written out in full (an array of bases with their PRG coordinates) and the rule is applied to the
 * finished array.
 *
 * Semantics restated ("PRG sketch"):
 *   local graph   every maximal run of bases between two markers is a node (possibly empty); a marker owns one space on
 *                 either side; odd marker m opens / closes site m, m+1 separates its alleles.
 *   k-mer nodes   forward-greedy, as pandora builds them.  Roots: on every walk from the graph start, the leftmost
 *                 minimum of the first w k-mers (of all k-mers if the walk holds fewer).  From a k-mer node a, along every
 *                 walk that continues a: the first of the next w-1 k-mers whose canonical hash is <= hash(a); if there
 *                 is none, the leftmost minimum of the next w k-mers; if the walk ends before w more k-mers exist, a is
 *                 joined to the sink.  Every k-mer reached this way is a node and is expanded in turn.
 *   ids           0 = source; k-mer nodes ordered by (first base coordinate, end coordinate of the last base, interval
 *                 list); last = sink.
 *   min path      number of edges on the shortest source -> sink path.
 * orc_kg_walk_check restates the *read-side* definition (orc_sketch's: all minima of every window of w consecutive k-mers)
 * over every walk fragment of the graph and reports which k-mer nodes it confirms and how many it finds that the greedy
 * construction missed (must be 0: a read minimizer on a PRG walk without its node would lose its hit).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#include "oracle_index.h"

/* ------------------------------------------------------------------------------------------------------------------
 * PRG string -> local graph
 * ---------------------------------------------------------------------------------------------------------------- */
static uint32_t lg_add_node(orc_kgraph* g, uint32_t a, uint32_t b)
{
    if (g->n_ln == g->cap_ln) {
        g->cap_ln = g->cap_ln ? 2 * g->cap_ln : 64;
        g->ln = (lnode*)realloc(g->ln, sizeof(lnode) * g->cap_ln);
    }
    lnode* n = &g->ln[g->n_ln];
    memset(n, 0, sizeof(*n));
    n->start = a;
    n->end = b < a ? a : b;
    return g->n_ln++;
}

static void lg_add_edge(orc_kgraph* g, uint32_t from, uint32_t to)
{
    lnode* n = &g->ln[from];
    if (n->n_out == n->cap_out) {
        n->cap_out = n->cap_out ? 2 * n->cap_out : 4;
        n->out = (uint32_t*)realloc(n->out, sizeof(uint32_t) * n->cap_out);
    }
    n->out[n->n_out++] = to;
}

typedef struct {
    int marker;
    uint32_t pre;
    uint32_t* ends;
    uint32_t n_ends, cap_ends;
} open_site;

static void site_add_end(open_site* s, uint32_t node)
{
    if (s->n_ends == s->cap_ends) {
        s->cap_ends = s->cap_ends ? 2 * s->cap_ends : 4;
        s->ends = (uint32_t*)realloc(s->ends, sizeof(uint32_t) * s->cap_ends);
    }
    s->ends[s->n_ends++] = node;
}

static int is_digit(char c) { return c >= '0' && c <= '9'; }

static void lg_parse(orc_kgraph* g)
{
    const char* s = g->s;
    const uint32_t n = g->len;
    open_site* stack = NULL;
    uint32_t depth = 0, cap_depth = 0;
    uint32_t* feed = (uint32_t*)malloc(sizeof(uint32_t) * 4); /* nodes that lead into the next segment */
    uint32_t n_feed = 0, cap_feed = 4;
    uint32_t seg_start = 0, i = 0;
    for (;;) {
        /* the segment runs up to the next marker (minus the space it owns) or to the end of the string */
        uint32_t j = i;
        while (j < n && !is_digit(s[j])) ++j;
        uint32_t seg_end = j;
        if (j < n && j > 0 && s[j - 1] == ' ') seg_end = j - 1;
        for (uint32_t q = seg_start; q < seg_end && q < n; ++q)
            if (s[q] == ' ') g->error = 1;
        uint32_t node = lg_add_node(g, seg_start, seg_end);
        for (uint32_t f = 0; f < n_feed; ++f) lg_add_edge(g, feed[f], node);
        if (j >= n) break;
        int m = 0;
        while (j < n && is_digit(s[j])) m = m * 10 + (s[j++] - '0');
        i = j;
        if (i < n && s[i] == ' ') ++i;
        seg_start = i;
        if (m & 1) {
            if (depth && stack[depth - 1].marker == m) { /* the site closes: all its allele ends lead on */
                open_site* st = &stack[depth - 1];
                site_add_end(st, node);
                if (st->n_ends > cap_feed) {
                    cap_feed = st->n_ends;
                    feed = (uint32_t*)realloc(feed, sizeof(uint32_t) * cap_feed);
                }
                memcpy(feed, st->ends, sizeof(uint32_t) * st->n_ends);
                n_feed = st->n_ends;
                if (st->n_ends < 2) g->error = 1;
                free(st->ends);
                --depth;
            } else { /* a site opens behind `node` */
                if (depth == cap_depth) {
                    cap_depth = cap_depth ? 2 * cap_depth : 8;
                    stack = (open_site*)realloc(stack, sizeof(open_site) * cap_depth);
                }
                memset(&stack[depth], 0, sizeof(open_site));
                stack[depth].marker = m;
                stack[depth].pre = node;
                ++depth;
                feed[0] = node;
                n_feed = 1;
            }
        } else { /* allele separator of the innermost open site */
            if (!depth || stack[depth - 1].marker != m - 1) {
                g->error = 1;
                break;
            }
            site_add_end(&stack[depth - 1], node);
            feed[0] = stack[depth - 1].pre;
            n_feed = 1;
        }
    }
    if (depth) {
        g->error = 1;
        for (uint32_t d = 0; d < depth; ++d) free(stack[d].ends);
    }
    free(stack);
    free(feed);
}

/* ------------------------------------------------------------------------------------------------------------------
 * walk fragments
 * ---------------------------------------------------------------------------------------------------------------- */
#define MAX_ITEMS 8192

typedef struct {
    item it[MAX_ITEMS];
    uint32_t n;       /* items */
    uint32_t n_bases; /* base items among them */
} fragment;

static int base_code(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

static char frag_base(const orc_kgraph* g, const item* it) { return g->s[g->ln[it->node].start + it->off]; }

/* canonical hash and strand of the k-mer whose first base is item `first` (a base item) of the fragment;
 * *last receives the item index of its k-th base.  Returns 0 if the fragment holds fewer than k bases from there. */
static int frag_kmer(const orc_kgraph* g, const fragment* f, uint32_t first, uint64_t* h, uint8_t* strand, uint32_t* last)
{
    const int k = g->k;
    uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1), fw = 0, rc = 0;
    int got = 0;
    uint32_t i = first;
    for (; i < f->n && got < k; ++i) {
        if (f->it[i].off == NO_OFF) continue;
        int c = base_code(frag_base(g, &f->it[i]));
        if (c > 3) return 0;
        fw = (fw << 2) | (uint64_t)c;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
        ++got;
    }
    if (got < k) return 0;
    uint64_t hf = orc_hash64(fw & mask, mask), hr = orc_hash64(rc & mask, mask);
    *h = hf < hr ? hf : hr;
    *strand = hf <= hr ? 1 : 0;
    *last = i - 1;
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------------
 * k-mer node set
 * ---------------------------------------------------------------------------------------------------------------- */
static uint32_t make_intervals(const orc_kgraph* g, const item* it, uint32_t n, uint32_t* iv)
{
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const lnode* ln = &g->ln[it[i].node];
        if (it[i].off == NO_OFF) {
            iv[2 * m] = ln->start;
            iv[2 * m + 1] = ln->start;
            ++m;
            continue;
        }
        uint32_t c = ln->start + it[i].off;
        if (m && i && it[i - 1].off != NO_OFF && it[i - 1].node == it[i].node && iv[2 * m - 1] == c) {
            iv[2 * m - 1] = c + 1;
        } else {
            iv[2 * m] = c;
            iv[2 * m + 1] = c + 1;
            ++m;
        }
    }
    return m;
}

static uint64_t iv_hash(const uint32_t* iv, uint32_t n)
{
    uint64_t h = 1469598103934665603ULL;
    for (uint32_t i = 0; i < 2 * n; ++i) {
        h ^= iv[i];
        h *= 1099511628211ULL;
    }
    return h;
}

static void table_insert(orc_kgraph* g, uint32_t idx)
{
    uint64_t h = iv_hash(g->kn[idx].iv, g->kn[idx].n_iv);
    uint32_t s = (uint32_t)(h & (g->table_size - 1));
    while (g->table[s]) s = (s + 1) & (g->table_size - 1);
    g->table[s] = idx + 1;
}

static uint32_t table_find(const orc_kgraph* g, const uint32_t* iv, uint32_t n_iv)
{
    if (!g->table_size) return 0xFFFFFFFFu;
    uint32_t s = (uint32_t)(iv_hash(iv, n_iv) & (g->table_size - 1));
    while (g->table[s]) {
        const knode* kn = &g->kn[g->table[s] - 1];
        if (kn->n_iv == n_iv && !memcmp(kn->iv, iv, sizeof(uint32_t) * 2 * n_iv)) return g->table[s] - 1;
        s = (s + 1) & (g->table_size - 1);
    }
    return 0xFFFFFFFFu;
}

/* index of the node for items [first, last] of the fragment; created (and queued: nodes are expanded in creation order) if new */
static uint32_t node_for(orc_kgraph* g, const fragment* f, uint32_t first, uint32_t last, uint64_t h, uint8_t strand)
{
    uint32_t iv[2 * 80];
    uint32_t n_items = last - first + 1;
    uint32_t* ivp = n_items <= 80 ? iv : (uint32_t*)malloc(sizeof(uint32_t) * 2 * n_items);
    uint32_t n_iv = make_intervals(g, &f->it[first], n_items, ivp);
    uint32_t found = table_find(g, ivp, n_iv);
    if (found == 0xFFFFFFFFu) {
        if (g->n_kn == g->cap_kn) {
            g->cap_kn = g->cap_kn ? 2 * g->cap_kn : 1024;
            g->kn = (knode*)realloc(g->kn, sizeof(knode) * g->cap_kn);
        }
        if (2 * (g->n_kn + 1) > g->table_size) {
            g->table_size = g->table_size ? 2 * g->table_size : 4096;
            free(g->table);
            g->table = (uint32_t*)calloc(g->table_size, sizeof(uint32_t));
            for (uint32_t i = 0; i < g->n_kn; ++i) table_insert(g, i);
        }
        knode* kn = &g->kn[g->n_kn];
        memset(kn, 0, sizeof(*kn));
        kn->n_iv = n_iv;
        kn->iv = (uint32_t*)malloc(sizeof(uint32_t) * 2 * n_iv);
        memcpy(kn->iv, ivp, sizeof(uint32_t) * 2 * n_iv);
        kn->n_items = n_items;
        kn->items = (item*)malloc(sizeof(item) * n_items);
        memcpy(kn->items, &f->it[first], sizeof(item) * n_items);
        kn->hash = h;
        kn->strand = strand;
        found = g->n_kn++;
        table_insert(g, found);
    }
    if (ivp != iv) free(ivp);
    return found;
}

static void add_edge(orc_kgraph* g, uint32_t from, uint32_t to)
{
    if (g->n_edges == g->cap_edges) {
        g->cap_edges = g->cap_edges ? 2 * g->cap_edges : 4096;
        g->edges = (kedge*)realloc(g->edges, sizeof(kedge) * g->cap_edges);
    }
    g->edges[g->n_edges].from = from;
    g->edges[g->n_edges].to = to;
    ++g->n_edges;
}

/* ------------------------------------------------------------------------------------------------------------------
 * enumeration of all ways to continue a fragment by `want` bases; `leaf` is called once per finished fragment with
 * ended = 1 if the walk reached the end of the graph before `want` bases could be added
 * ---------------------------------------------------------------------------------------------------------------- */
typedef void (*leaf_fn)(orc_kgraph* g, fragment* f, int ended, void* arg);

static void extend(orc_kgraph* g, fragment* f, uint32_t want, leaf_fn leaf, void* arg);

/* continue through the successors of local node `node` (all of whose bases are used) */
static void extend_out(orc_kgraph* g, fragment* f, uint32_t node, uint32_t want, leaf_fn leaf, void* arg)
{
    const lnode* ln = &g->ln[node];
    if (ln->n_out == 0) { /* the end of the graph */
        leaf(g, f, 1, arg);
        return;
    }
    for (uint32_t o = 0; o < ln->n_out; ++o) {
        uint32_t nx = ln->out[o];
        if (f->n >= MAX_ITEMS) {
            g->error = 1;
            return;
        }
        if (g->ln[nx].end > g->ln[nx].start) {
            f->it[f->n].node = nx;
            f->it[f->n].off = 0;
            ++f->n;
            ++f->n_bases;
            extend(g, f, want - 1, leaf, arg);
            --f->n;
            --f->n_bases;
        } else {
            f->it[f->n].node = nx;
            f->it[f->n].off = NO_OFF;
            ++f->n;
            extend_out(g, f, nx, want, leaf, arg);
            --f->n;
        }
    }
}

static void extend(orc_kgraph* g, fragment* f, uint32_t want, leaf_fn leaf, void* arg)
{
    if (want == 0) {
        leaf(g, f, 0, arg);
        return;
    }
    const item last = f->it[f->n - 1]; /* always a base item here */
    const lnode* ln = &g->ln[last.node];
    if (ln->start + last.off + 1 < ln->end) {
        if (f->n >= MAX_ITEMS) {
            g->error = 1;
            return;
        }
        f->it[f->n].node = last.node;
        f->it[f->n].off = last.off + 1;
        ++f->n;
        ++f->n_bases;
        extend(g, f, want - 1, leaf, arg);
        --f->n;
        --f->n_bases;
    } else {
        extend_out(g, f, last.node, want, leaf, arg);
    }
}

/* positions (item indexes) of the base items of a fragment */
static uint32_t base_positions(const fragment* f, uint32_t* pos)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < f->n; ++i)
        if (f->it[i].off != NO_OFF) pos[n++] = i;
    return n;
}

/* fragments that end in crossed empty nodes carry them as trailing items; a k-mer never includes them (node_for is
 * given [first base, last base]) */

static void root_leaf(orc_kgraph* g, fragment* f, int ended, void* arg)
{
    (void)ended;
    (void)arg;
    uint32_t pos[MAX_ITEMS];
    uint32_t nb = base_positions(f, pos);
    if ((int)nb < g->k) return; /* this walk holds no k-mer */
    uint32_t nk = nb - (uint32_t)g->k + 1;
    uint64_t best_h = 0;
    uint8_t best_s = 0;
    uint32_t best = 0xFFFFFFFFu, best_last = 0;
    for (uint32_t j = 0; j < nk; ++j) {
        uint64_t h;
        uint8_t st;
        uint32_t last;
        if (!frag_kmer(g, f, pos[j], &h, &st, &last)) {
            g->error = 1;
            return;
        }
        if (best == 0xFFFFFFFFu || h < best_h) {
            best = j;
            best_h = h;
            best_s = st;
            best_last = last;
        }
    }
    add_edge(g, SRC, node_for(g, f, pos[best], best_last, best_h, best_s));
}

typedef struct {
    uint32_t from;      /* the node being expanded */
    uint32_t from_bases; /* = k */
} grow_arg;

static void grow_leaf(orc_kgraph* g, fragment* f, int ended, void* arg)
{
    const grow_arg* ga = (const grow_arg*)arg;
    const uint64_t from_hash = g->kn[ga->from].hash;
    uint32_t pos[MAX_ITEMS];
    uint32_t nb = base_positions(f, pos);
    uint32_t m = nb - (uint32_t)g->k; /* later k-mers on this fragment: shifts 1..m of the node's k-mer */
    uint64_t best_h = 0;
    uint8_t best_s = 0;
    uint32_t best = 0, best_last = 0;
    for (uint32_t j = 1; j <= m; ++j) {
        uint64_t h;
        uint8_t st;
        uint32_t last;
        if (!frag_kmer(g, f, pos[j], &h, &st, &last)) {
            g->error = 1;
            return;
        }
        if (h <= from_hash) { /* the first later k-mer that is not larger: the next minimizer on this walk */
            add_edge(g, ga->from, node_for(g, f, pos[j], last, h, st));
            return;
        }
        if (best == 0 || h < best_h) {
            best = j;
            best_h = h;
            best_s = st;
            best_last = last;
        }
    }
    if (!ended && (int)m == g->w) { /* none: the leftmost minimum of the w k-mers that follow */
        add_edge(g, ga->from, node_for(g, f, pos[best], best_last, best_h, best_s));
        return;
    }
    add_edge(g, ga->from, SNK); /* the walk ends inside the next window */
}

/* every fragment of `want` bases that starts at the graph's first base(s) */
static void from_graph_start(orc_kgraph* g, fragment* f, uint32_t want, leaf_fn leaf, void* arg)
{
    f->n = 0;
    f->n_bases = 0;
    if (g->ln[0].end > g->ln[0].start) {
        f->it[0].node = 0;
        f->it[0].off = 0;
        f->n = 1;
        f->n_bases = 1;
        extend(g, f, want - 1, leaf, arg);
    } else {
        extend_out(g, f, 0, want, leaf, arg);
    }
}

/* drop leading empty items so that a fragment starts with a base */
static void strip_leading_empties(fragment* f)
{
    uint32_t i = 0;
    while (i < f->n && f->it[i].off == NO_OFF) ++i;
    if (i) {
        memmove(f->it, f->it + i, sizeof(item) * (f->n - i));
        f->n -= i;
    }
}

static void root_leaf_stripped(orc_kgraph* g, fragment* f, int ended, void* arg)
{
    fragment* c = (fragment*)malloc(sizeof(fragment));
    memcpy(c, f, sizeof(fragment));
    strip_leading_empties(c);
    root_leaf(g, c, ended, arg);
    free(c);
}

static int edge_cmp(const void* a, const void* b)
{
    const kedge *x = (const kedge*)a, *y = (const kedge*)b;
    if (x->from != y->from) return x->from < y->from ? -1 : 1;
    if (x->to != y->to) return x->to < y->to ? -1 : 1;
    return 0;
}

static const orc_kgraph* g_sort_ctx;

static int node_order_cmp(const void* a, const void* b)
{
    const knode *x = &g_sort_ctx->kn[*(const uint32_t*)a], *y = &g_sort_ctx->kn[*(const uint32_t*)b];
    if (x->iv[0] != y->iv[0]) return x->iv[0] < y->iv[0] ? -1 : 1;
    uint32_t xe = x->iv[2 * x->n_iv - 1], ye = y->iv[2 * y->n_iv - 1];
    if (xe != ye) return xe < ye ? -1 : 1;
    uint32_t n = x->n_iv < y->n_iv ? x->n_iv : y->n_iv;
    for (uint32_t i = 0; i < 2 * n; ++i)
        if (x->iv[i] != y->iv[i]) return x->iv[i] < y->iv[i] ? -1 : 1;
    if (x->n_iv != y->n_iv) return x->n_iv < y->n_iv ? -1 : 1;
    return 0;
}

static uint32_t final_id(const orc_kgraph* g, uint32_t ref)
{
    if (ref == SRC) return 0;
    if (ref == SNK) return g->n_kn + 1;
    return g->kn[ref].id;
}

ORC_API void orc_kg_free(orc_kgraph* g)
{
    if (!g) return;
    for (uint32_t i = 0; i < g->n_ln; ++i) free(g->ln[i].out);
    for (uint32_t i = 0; i < g->n_kn; ++i) {
        free(g->kn[i].iv);
        free(g->kn[i].items);
    }
    free(g->ln); free(g->kn); free(g->table); free(g->edges); free(g->order); free(g->s);
    free(g);
}

/*
 * Sketch one PRG string.  Returns NULL for a malformed PRG.
 */
ORC_API orc_kgraph* orc_index_prg(const char* prg, int w, int k)
{
    if (k < 1 || k > 32 || w < 1) return NULL;
    orc_kgraph* g = (orc_kgraph*)calloc(1, sizeof(orc_kgraph));
    g->len = (uint32_t)strlen(prg);
    g->s = (char*)malloc(g->len + 1);
    memcpy(g->s, prg, g->len + 1);
    g->w = w;
    g->k = k;
    lg_parse(g);
    if (g->error) {
        orc_kg_free(g);
        return NULL;
    }
    fragment* f = (fragment*)malloc(sizeof(fragment));
    /* roots: the leftmost minimum of the first min(w, all) k-mers of every walk */
    from_graph_start(g, f, (uint32_t)(w + k - 1), root_leaf_stripped, NULL);
    /* expansion in creation order */
    for (uint32_t a = 0; a < g->n_kn && !g->error; ++a) {
        f->n = g->kn[a].n_items;
        f->n_bases = (uint32_t)k;
        memcpy(f->it, g->kn[a].items, sizeof(item) * f->n);
        grow_arg ga = { a, (uint32_t)k };
        extend(g, f, (uint32_t)w, grow_leaf, &ga);
    }
    free(f);
    if (g->error) {
        orc_kg_free(g);
        return NULL;
    }
    if (g->n_kn == 0) add_edge(g, SRC, SNK); /* no walk holds a k-mer */
    /* ids */
    g->order = (uint32_t*)malloc(sizeof(uint32_t) * (g->n_kn ? g->n_kn : 1));
    for (uint32_t i = 0; i < g->n_kn; ++i) g->order[i] = i;
    g_sort_ctx = g;
    qsort(g->order, g->n_kn, sizeof(uint32_t), node_order_cmp);
    for (uint32_t i = 0; i < g->n_kn; ++i) g->kn[g->order[i]].id = i + 1;
    /* distinct edges in final ids */
    for (uint64_t e = 0; e < g->n_edges; ++e) {
        g->edges[e].from = final_id(g, g->edges[e].from);
        g->edges[e].to = final_id(g, g->edges[e].to);
    }
    qsort(g->edges, g->n_edges, sizeof(kedge), edge_cmp);
    uint64_t m = 0;
    for (uint64_t e = 0; e < g->n_edges; ++e)
        if (m == 0 || edge_cmp(&g->edges[e], &g->edges[m - 1]) != 0) g->edges[m++] = g->edges[e];
    g->n_edges = m;
    /* shortest source -> sink path, in edges: breadth-first over the (from-sorted) edge list */
    uint32_t n_nodes = g->n_kn + 2;
    uint64_t* first_edge = (uint64_t*)calloc(n_nodes + 1, sizeof(uint64_t));
    for (uint64_t e = 0; e < g->n_edges; ++e) first_edge[g->edges[e].from + 1]++;
    for (uint32_t i = 0; i < n_nodes; ++i) first_edge[i + 1] += first_edge[i];
    uint32_t* dist = (uint32_t*)malloc(sizeof(uint32_t) * n_nodes);
    uint32_t* queue = (uint32_t*)malloc(sizeof(uint32_t) * n_nodes);
    memset(dist, 0xFF, sizeof(uint32_t) * n_nodes);
    uint32_t qh = 0, qt = 0;
    dist[0] = 0;
    queue[qt++] = 0;
    while (qh < qt) {
        uint32_t u = queue[qh++];
        for (uint64_t e = first_edge[u]; e < first_edge[u + 1]; ++e) {
            uint32_t v = g->edges[e].to;
            if (dist[v] == 0xFFFFFFFFu) {
                dist[v] = dist[u] + 1;
                queue[qt++] = v;
            }
        }
    }
    g->min_path_len = dist[n_nodes - 1] == 0xFFFFFFFFu ? 0 : dist[n_nodes - 1];
    free(first_edge); free(dist); free(queue);
    return g;
}

/* ------------------------------------------------------------------------------------------------------------------
 * accessors (ids: 0 = source, 1..n = k-mer nodes, n+1 = sink)
 * ---------------------------------------------------------------------------------------------------------------- */
ORC_API uint32_t orc_kg_n_nodes(const orc_kgraph* g) { return g->n_kn + 2; }
ORC_API uint32_t orc_kg_min_path_len(const orc_kgraph* g) { return g->min_path_len; }
ORC_API uint64_t orc_kg_n_edges(const orc_kgraph* g) { return g->n_edges; }
ORC_API uint32_t orc_kg_n_local_nodes(const orc_kgraph* g) { return g->n_ln; }

ORC_API void orc_kg_local_nodes(const orc_kgraph* g, uint32_t* starts, uint32_t* ends)
{
    for (uint32_t i = 0; i < g->n_ln; ++i) {
        starts[i] = g->ln[i].start;
        ends[i] = g->ln[i].end;
    }
}

/* hash[i], strand[i] of k-mer node id i + 1 */
ORC_API void orc_kg_kmers(const orc_kgraph* g, uint64_t* hash, uint8_t* strand)
{
    for (uint32_t i = 0; i < g->n_kn; ++i) {
        hash[i] = g->kn[g->order[i]].hash;
        strand[i] = g->kn[g->order[i]].strand;
    }
}

ORC_API void orc_kg_edges(const orc_kgraph* g, uint32_t* from, uint32_t* to)
{
    for (uint64_t e = 0; e < g->n_edges; ++e) {
        from[e] = g->edges[e].from;
        to[e] = g->edges[e].to;
    }
}

/* interval list of k-mer node `id` (1..n): writes up to cap (start, end) pairs, returns their number */
ORC_API uint32_t orc_kg_path(const orc_kgraph* g, uint32_t id, uint32_t* iv, uint32_t cap)
{
    if (id < 1 || id > g->n_kn) return 0;
    const knode* kn = &g->kn[g->order[id - 1]];
    for (uint32_t i = 0; i < kn->n_iv && i < cap; ++i) {
        iv[2 * i] = kn->iv[2 * i];
        iv[2 * i + 1] = kn->iv[2 * i + 1];
    }
    return kn->n_iv;
}

/* per-base coverage of the local nodes that the k-mers path[0..n_path) (node ids 1..n) run through, in the order the path first
 * reaches them: base = the largest covg_total[id] among the path's k-mers that cover it, 0 if none does (pandora
 * get_covgs_along_localnode_path [UPSTREAM-MEMORY]).  covg_total is indexed by node id (n_nodes entries).  Returns the number of
 * bases, the first `cap` of them in out[]. */
ORC_API int64_t orc_kg_base_coverage(const orc_kgraph* g, const uint32_t* path, int64_t n_path, const uint32_t* covg_total, uint32_t* out, int64_t cap)
{
    int64_t* first_base = (int64_t*)malloc(sizeof(int64_t) * (g->n_ln ? g->n_ln : 1)); /* position of a local node's first base in out[], -1 = not reached yet */
    for (uint32_t i = 0; i < g->n_ln; ++i) first_base[i] = -1;
    int64_t n_out = 0;
    for (int64_t pi = 0; pi < n_path; ++pi) {
        const uint32_t id = path[pi];
        if (id < 1 || id > g->n_kn) continue;
        const knode* kn = &g->kn[g->order[id - 1]];
        for (uint32_t j = 0; j < kn->n_iv; ++j) {
            const uint32_t a = kn->iv[2 * j], b = kn->iv[2 * j + 1];
            uint32_t ln = g->n_ln;
            for (uint32_t q = 0; q < g->n_ln; ++q) /* the local node this interval lies in (an empty node for an empty interval) */
                if (g->ln[q].start <= a && b <= g->ln[q].end && (a < b || g->ln[q].start == g->ln[q].end)) {
                    ln = q;
                    break;
                }
            if (ln == g->n_ln) continue;
            if (first_base[ln] < 0) {
                first_base[ln] = n_out;
                for (uint32_t x = g->ln[ln].start; x < g->ln[ln].end; ++x, ++n_out)
                    if (n_out < cap) out[n_out] = 0;
            }
            for (uint32_t x = a; x < b; ++x) {
                const int64_t at = first_base[ln] + (int64_t)(x - g->ln[ln].start);
                if (at < cap && out[at] < covg_total[id]) out[at] = covg_total[id];
            }
        }
    }
    free(first_base);
    return n_out;
}

/* ------------------------------------------------------------------------------------------------------------------
 * the read-side definition over every walk fragment: all minima of every window of w consecutive k-mers (and, for a
 * complete walk that holds fewer than w k-mers, of the whole walk)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t windows, missing;
    int whole_walks_only; /* fragments from the graph start: count the short complete walks only */
} walk_arg;

static void walk_leaf(orc_kgraph* g, fragment* f0, int ended, void* arg)
{
    walk_arg* wa = (walk_arg*)arg;
    fragment* f = f0;
    fragment* copy = NULL;
    if (f->n && f->it[0].off == NO_OFF) {
        copy = (fragment*)malloc(sizeof(fragment));
        memcpy(copy, f0, sizeof(fragment));
        strip_leading_empties(copy);
        f = copy;
    }
    uint32_t pos[MAX_ITEMS];
    uint32_t nb = base_positions(f, pos);
    if (ended && !wa->whole_walks_only) goto done;        /* a shorter window is only a window if it is the whole walk */
    if (!ended && wa->whole_walks_only) goto done;        /* full windows are enumerated from every start position instead */
    if ((int)nb < g->k) goto done;
    {
        uint32_t nk = nb - (uint32_t)g->k + 1;
        uint64_t h[4096];
        uint8_t st[4096];
        uint32_t last[4096];
        uint64_t m = ~0ULL;
        if (nk > 4096) {
            g->error = 1;
            goto done;
        }
        for (uint32_t j = 0; j < nk; ++j) {
            if (!frag_kmer(g, f, pos[j], &h[j], &st[j], &last[j])) {
                g->error = 1;
                goto done;
            }
            if (h[j] < m) m = h[j];
        }
        ++wa->windows;
        for (uint32_t j = 0; j < nk; ++j) {
            if (h[j] != m) continue;
            uint32_t iv[2 * 80];
            uint32_t n_items = last[j] - pos[j] + 1;
            if (n_items > 80) {
                g->error = 1;
                goto done;
            }
            uint32_t n_iv = make_intervals(g, &f->it[pos[j]], n_items, iv);
            uint32_t idx = table_find(g, iv, n_iv);
            if (idx == 0xFFFFFFFFu) ++wa->missing;
            else g->kn[idx].walk_confirmed = 1;
        }
    }
done:
    free(copy);
}

/*
 * out[0] = windows examined, out[1] = window minimizers that are NOT k-mer nodes (must be 0),
 * out[2] = k-mer nodes that are a window minimizer on some walk, out[3] = k-mer nodes that are not (the forward-greedy
 * construction continues a node along every walk, also along walks on which that node is not itself a minimizer).
 * confirmed (may be NULL) receives one byte per k-mer node id 1..n.
 */
ORC_API int orc_kg_walk_check(orc_kgraph* g, uint64_t out[4], uint8_t* confirmed)
{
    fragment* f = (fragment*)malloc(sizeof(fragment));
    walk_arg wa = { 0, 0, 0 };
    const uint32_t want = (uint32_t)(g->w + g->k - 1);
    for (uint32_t i = 0; i < g->n_kn; ++i) g->kn[i].walk_confirmed = 0;
    for (uint32_t n = 0; n < g->n_ln; ++n)
        for (uint32_t off = 0; g->ln[n].start + off < g->ln[n].end; ++off) {
            f->it[0].node = n;
            f->it[0].off = off;
            f->n = 1;
            f->n_bases = 1;
            extend(g, f, want - 1, walk_leaf, &wa);
        }
    wa.whole_walks_only = 1;
    from_graph_start(g, f, want, walk_leaf, &wa);
    free(f);
    out[0] = wa.windows;
    out[1] = wa.missing;
    out[2] = out[3] = 0;
    for (uint32_t i = 0; i < g->n_kn; ++i) {
        const knode* kn = &g->kn[g->order[i]];
        if (kn->walk_confirmed) ++out[2]; else ++out[3];
        if (confirmed) confirmed[i] = kn->walk_confirmed;
    }
    return g->error ? -1 : 0;
}
