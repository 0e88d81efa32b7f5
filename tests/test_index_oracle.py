"""Row a-4 (`pandora index`, /root/reference/src/lib.rs:479-510): the product's index builder (csrc/prg.cpp, kmergraph.cpp,
index.cpp) against the oracle's own, independently written one (oracle/oracle_index.c).

Equality, not containment: keys, records per key, strand flags, node numbering, per-PRG shortest k-mer path, and -- through
the files `pandora index` writes -- every k-mer node's interval list and every edge of every k-mer graph.  The PRG syntax
and the interval convention are the part the reference pins (tests/golden/prg_syntax/dr.prg = /root/reference/tests/cases/
expected/dr.prg; denovo_paths_example.txt = /root/reference/src/lib.rs:3010-3038)."""
import os
import re

import numpy as np
import pytest

from util import GOLDEN


def _read_prg_file(path):
    names, prgs = [], []
    for line in open(path):
        if line.startswith(">"):
            names.append(line[1:].split()[0])
        elif line.strip("\n"):
            prgs.append(line.rstrip("\n"))
    return names, prgs


def _product_index(tmp_path, names, prgs, w, k, threads=4):
    from drprg_amd import Context
    prg = str(tmp_path / "dr.prg")
    with open(prg, "w") as fh:
        fh.write("".join(f">{n}\n{p}\n" for n, p in zip(names, prgs)))
    ctx = Context(prg, w, k, device=-1, from_files=False, threads=threads)
    return ctx, prg


def _assert_same_index(a, b):
    for key in ("knode_base", "min_path_len", "keys", "rec_off", "rec_prg", "rec_knode", "rec_strand"):
        assert a[key].shape == b[key].shape, key
        assert np.array_equal(a[key], b[key]), key


EDGE_PRGS = {
    "ends_with_site": "ACGTACGTTTGACCAGTAGGACCATTAGACCAGATTACAGGATC 5 GG 6 AG 5 ",
    "ends_with_empty_allele": "ACGTACGTTTGACCAGTAGGACCATTAGACCAGATTACAGGATC 5 G 6  5 ",
    "long_or_empty_allele_at_the_end": "ACGTACGTTTGACCAGTAGGACCATTAGACCAGATTACAGGATC 5 GTTAGACAGGAT 6  5 ",
    "starts_with_site": " 5 A 6 C 5 ACGTACGTTTGACCAGTAGGACCATTAGACCAGATTACAGGATCAGGT",
    "starts_with_empty_allele": " 5  6 C 5 ACGTACGTTTGACCAGTAGGACCATTAGACCAGATTACAGGATCAGGT",
    "fewer_than_w_kmers": "ACGTACGTAGGATCCA",
    "short_with_site": "ACGTACG 5 T 6 G 5 AGGATCCAG",
    "shorter_than_k": "ACGT",
    "adjacent_sites": "ACGTACGTTTGACCAGTAGG 5 A 6 C 5  7 G 8 T 7 ACCATTAGACCAGATTACAGGATCAGGTAGGCAT",
    "nested_with_empty_first_segment": "ACGTACGTTTGACCAGTAGGACC 5  7 G 8 T 7 TCACGG 6 TTGGGCGGCAGCGACGCT 5 ATTAGACCAGATTACAGGATCAGGTAGGCATCAGGAT",
    "three_levels": "ACGTACGTTTGACCAGTAGGACC 5 AC 7 G 9 T 10 TT 9 A 8 T 7 TCACGG 6 TTGGGC 5 ATTAGACCAGATTACAGGATCAGGTAGGCATCAGGAT",
    "lower_case": "acgtacgtttgaccagtaggaccattagaccagattacaggatc 5 g 6 t 5 aggtaccagatagacagat",
}
WK = [(11, 15), (14, 15), (3, 5), (1, 7), (5, 9), (19, 21), (11, 31)]


@pytest.mark.parametrize("name", sorted(EDGE_PRGS))
def test_edge_case_prgs(tmp_path, oracle, name):
    for w, k in WK:
        ctx, _ = _product_index(tmp_path, [name], [EDGE_PRGS[name]], w, k)
        _assert_same_index(ctx.export_index(), oracle.build_index([EDGE_PRGS[name]], w, k))
        ctx.close()


@pytest.mark.parametrize("w,k", WK)
def test_reference_dr_prg(tmp_path, oracle, w, k):
    """the PRG file the reference's build test expects (two loci, nested sites, empty alleles, a PRG ending in a site)"""
    names, prgs = _read_prg_file(os.path.join(GOLDEN, "prg_syntax", "dr.prg"))
    ctx, _ = _product_index(tmp_path, names, prgs, w, k)
    _assert_same_index(ctx.export_index(), oracle.build_index(prgs, w, k))


def test_local_graph_ids_and_intervals_match_the_reference_example(oracle):
    """The denovo_paths.txt embedded in /root/reference/src/lib.rs:3010-3038 lists local-graph nodes of the real mtb index
    as (id [start, end) sequence): ids count the segments in PRG-string order, intervals are offsets into the PRG string
    with markers and their spaces included.  Rebuild the PRG prefix those lines imply (unlisted alleles filled with one
    base) and check the oracle's parser gives every listed node its id, interval and sequence."""
    want, gid = {}, False
    for line in open(os.path.join(GOLDEN, "denovo_paths_example.txt")):
        if line.strip() in ("gid", "ahpC"):
            gid = line.strip() == "gid"
        m = re.match(r"\((\d+) \[(\d+), (\d+)\) ([ACGT]*)\)", line)
        if m and gid:
            want[int(m.group(1))] = (int(m.group(2)), int(m.group(3)), m.group(4))
    assert sorted(want) == [0, 1, 3, 4, 6, 8, 10, 11]
    s = want[0][2] + " 5 C 6 T 5 " + want[3][2] + " 7 C 8 T 7 " + want[6][2] + " 9 " + "A" * 24 + " 10 " + want[8][2] \
        + " 10 G 9 " + want[10][2] + " 11 GC 12 G 11 " + "ACGTACGTACGTACGTACGTACGTACGTACGT"
    g = oracle.sketch_prg(s, 11, 15)
    for node, (a, b, seq) in want.items():
        assert (int(g["local_starts"][node]), int(g["local_ends"][node])) == (a, b), node
        assert s[a:b] == seq
    # an empty allele is the position between the two spaces (tests/golden/prg_syntax/dr.prg writes them as a double space)
    g = oracle.sketch_prg("ACGT 5  6 C 5 ACGTACGTACGTACGTACGTACGTACGT", 3, 5)
    assert list(zip(g["local_starts"].tolist(), g["local_ends"].tolist()))[:4] == [(0, 4), (7, 7), (10, 11), (14, 42)]


def _panels():
    from drprg_amd import synth
    yield "small_nested_indel", synth.small_panel(seed=7), (11, 15)
    yield "small_w5_k9", synth.small_panel(seed=3), (5, 9)
    yield "small_w16_k13", synth.small_panel(seed=5, n_loci=6, length=900, site_every=25), (16, 13)
    yield "mtb_like", synth.mtb_like_panel(), (11, 15)
    panel, _ = synth.panel_from_index_dir(os.path.join(GOLDEN, "downstream"))
    yield "reference_genes_fa_with_panel_bcf_sites", panel, (11, 15)
    yield "survey_8d_mtb_index", synth.mtb_8d_panel(), (11, 15)
    yield "big_first_40_loci", synth.Panel(*(lambda p: (p.names[:40], p.trees[:40]))(synth.big_panel())), (11, 15)


@pytest.mark.parametrize("label,panel,wk", list(_panels()), ids=[p[0] for p in _panels()])
def test_panels_equal_the_oracle_index(tmp_path, oracle, label, panel, wk):
    w, k = wk
    ctx, _ = _product_index(tmp_path, panel.names, panel.prgs, w, k)
    idx = ctx.export_index()
    _assert_same_index(idx, oracle.build_index(panel.prgs, w, k))
    assert len(idx["keys"]) > 100


def _parse_gfa(path):
    nodes, edges = [], []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if f[0] == "S":
            assert int(f[1]) == len(nodes)
            nodes.append([(int(a), int(b)) for a, b in re.findall(r"\[(\d+), (\d+)\)", f[2])])
        elif f[0] == "L":
            edges.append((int(f[1]), int(f[3])))
    return nodes, sorted(edges)


def test_written_kmer_graphs_equal_the_oracle_graphs(tmp_path, oracle):
    """`pandora index` output files: every S line's interval list and every L line of kmer_prgs/*.gfa, and the .idx records"""
    from drprg_amd import Pandora, synth
    from drprg_amd._lib import lib
    names, prgs = _read_prg_file(os.path.join(GOLDEN, "prg_syntax", "dr.prg"))
    panel = synth.small_panel(seed=21, n_loci=3, length=600, site_every=30)
    names, prgs = names + panel.names, prgs + panel.prgs
    prg = str(tmp_path / "dr.prg")
    with open(prg, "w") as fh:
        fh.write("".join(f">{n}\n{p}\n" for n, p in zip(names, prgs)))
    w, k = 11, 15
    assert lib.drprg_hip_index(prg.encode(), w, k, 2) == 0
    graph_paths = []
    for n, p in zip(names, prgs):
        nodes, edges = _parse_gfa(str(tmp_path / "kmer_prgs" / f"{n}.k{k}.w{w}.gfa"))
        g = oracle.sketch_prg(p, w, k, paths=True)
        graph_paths.append(g["paths"])
        assert len(nodes) == g["n_nodes"]
        assert nodes[0] == [] and nodes[-1] == []
        assert nodes[1:-1] == g["paths"]
        assert edges == sorted(map(tuple, g["edges"].tolist()))
    # the .idx: key, count, (prg, local node, strand) ...
    idx = oracle.build_index(prgs, w, k)
    lines = open(prg + f".k{k}.w{w}.idx").read().splitlines()
    assert int(lines[0]) == len(idx["keys"])
    for i, line in enumerate(lines[1:]):
        f = line.split("\t")
        assert int(f[0]) == int(idx["keys"][i])
        lo, hi = int(idx["rec_off"][i]), int(idx["rec_off"][i + 1])
        assert int(f[1]) == hi - lo
        for j, rec in enumerate(f[2:]):
            m = re.fullmatch(r"\((\d+), (\d+\{(?:\[\d+, \d+\))+\}), (\d+), ([01])\)", rec)  # (prg_id, path, knode_id, strand)
            assert m, rec
            p, node, strand = int(m.group(1)), int(m.group(3)), int(m.group(4))
            assert p == idx["rec_prg"][lo + j] and strand == idx["rec_strand"][lo + j]
            assert node == idx["rec_knode"][lo + j] - idx["knode_base"][p]
            assert [(int(a), int(b)) for a, b in re.findall(r"\[(\d+), (\d+)\)", m.group(2))] == graph_paths[p][node - 1]


def test_every_read_side_minimizer_of_a_prg_walk_is_a_node(oracle):
    """The read sketch (a-5) and the PRG sketch (a-4) must agree on PRG walks: every minimum of every window of w k-mers on
    every walk fragment is a k-mer node (otherwise an error-free read would lose that hit).  The converse does not hold
    and is not required: the forward-greedy construction (pandora's) continues a node along every walk, also walks on
    which that node is no minimizer, and so creates a few nodes no error-free read can hit."""
    from drprg_amd import synth
    names, prgs = _read_prg_file(os.path.join(GOLDEN, "prg_syntax", "dr.prg"))
    prgs = prgs + synth.small_panel(seed=7).prgs + synth.mtb_like_panel().prgs[:6] + list(EDGE_PRGS.values())
    confirmed = unconfirmed = 0
    for w, k in [(11, 15), (5, 9), (3, 5), (1, 7)]:
        for p in prgs:
            g = oracle.sketch_prg(p, w, k, walk_check=True)
            assert g["walk"]["missing"] == 0
            confirmed += g["walk"]["confirmed"]
            unconfirmed += g["walk"]["unconfirmed"]
    assert confirmed > 20 * unconfirmed  # (the extra nodes are a small minority)


def test_linear_prg_equals_the_read_sketch(oracle):
    """on a PRG without sites the k-mer nodes are exactly orc_sketch's minimizers of that sequence (a-4 == a-5)"""
    from drprg_amd import synth
    rng = np.random.default_rng(3)
    for w, k in WK:
        seq = synth.random_seq(rng, 700)
        g = oracle.sketch_prg(seq, w, k, paths=True)
        h, p, s = oracle.sketch(seq, w, k)
        assert [pp[0][0] for pp in g["paths"]] == p.tolist()
        assert np.array_equal(g["hash"], h) and np.array_equal(g["strand"], s)
        assert g["min_path_len"] == len(h) + 1


def test_malformed_prgs_are_refused_by_both(tmp_path, oracle):
    from drprg_amd import DependencyError
    for bad in ("ACGT 5 A 6 C", "ACGT 6 A 5 C 5 ", "ACGT 5 A 5 ACGT", "ACGT 5 A 6 C 7 ACGT"):
        with pytest.raises(ValueError):
            oracle.sketch_prg(bad, 11, 15)
        with pytest.raises(DependencyError):
            _product_index(tmp_path, ["bad"], [bad], 11, 15)
