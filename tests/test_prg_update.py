"""The PRG update (what MakePrg::update does with mafft + make_prg in the reference, /root/reference/src/lib.rs:279-456) on hand-made
denovo_paths.txt files: variants inside a node, at a node's edges, running into a site, across a site, across a nested site, inside a
nested allele, at both ends of a locus, insertions at a site boundary, deletions of whole sites, several variants whose stretches
overlap.  Checked every time: the updated PRG parses the way pandora parses (tests/util.py prg_language: markers in parse order), it
spells everything the old PRG spelled, and it spells the called path with the variants applied -- which the old one did not."""
import itertools
import os
import re

import pytest

from util import prg_language


def path_nodes(prg, choose):
    """the walk through `prg` that takes allele choose(site number in parse order) at every site: [(start, end, seq)] without empty nodes"""
    nodes = []
    counter = [0]

    def walk(lo, hi):  # the interval [lo, hi) of the string, at one nesting level
        i = lo
        while i < hi:
            m = re.compile(r" (\d+) ").search(prg, i, hi)
            if not m:
                if hi > i:
                    nodes.append((i, hi, prg[i:hi]))
                return
            if m.start() > i:
                nodes.append((i, m.start(), prg[i:m.start()]))
            num = int(m.group(1))
            assert num % 2 == 1
            site = counter[0]
            counter[0] += 1
            close = prg.index(" %d " % num, m.end() - 1)
            # the alleles: split at this site's separator at this level only
            inner_lo, inner_hi = m.end(), close
            cuts, depth_open, j = [inner_lo], [], inner_lo
            for t in re.finditer(r" (\d+) ", prg[inner_lo - 1:inner_hi + 1]):
                pass
            # scan tokens by hand so that shared spaces (" 6  5 ") are handled
            pos = inner_lo
            alleles = []
            start = inner_lo
            stack = []
            k = inner_lo - 1  # the space that closed the opening marker
            while k < inner_hi + 1:
                t = re.compile(r" (\d+) ").match(prg, k)
                if t and k >= inner_lo - 1:
                    v = int(t.group(1))
                    if not stack and v == num + 1:
                        alleles.append((start, k))
                        start = t.end()
                        k = t.end() - 1
                        continue
                    if v % 2 == 1:
                        if stack and stack[-1] == v:
                            stack.pop()
                        elif v != num:
                            stack.append(v)
                    k = t.end() - 1
                    continue
                k += 1
            alleles.append((start, max(start, inner_hi)))
            a = choose(site) % len(alleles)
            # sites inside the alleles not taken still consume their numbers in parse order
            for n, (alo, ahi) in enumerate(alleles):
                if n == a:
                    walk(alo, ahi)
                else:
                    counter[0] += len(set(int(x) for x in re.findall(r" (\d+) ", prg[max(alo - 1, 0):ahi + 1]) if int(x) % 2 == 1))
            i = close + len(" %d " % num)
        return

    walk(0, len(prg))
    return [n for n in nodes if n[1] > n[0]]


def write_paths(path, loci):
    """loci: [(name, nodes, [(pos1, ref, alt)])] in the layout of /root/reference/src/lib.rs:3010-3038"""
    with open(path, "w") as o:
        o.write("1 samples\nSample s\n%d loci with denovo variants\n" % len(loci))
        for name, nodes, variants in loci:
            o.write("%s\n%d nodes\n" % (name, len(nodes)))
            for i, (a, b, seq) in enumerate(nodes):
                o.write("(%d [%d, %d) %s)\n" % (i, a, b, seq))
            o.write("%d denovo variants for this locus\n" % len(variants))
            for pos1, ref, alt in variants:
                o.write("%d\t%s\t%s\n" % (pos1, ref, alt))


def apply(seq, variants):
    for pos1, ref, alt in sorted(variants, reverse=True):
        assert seq[pos1 - 1:pos1 - 1 + len(ref)] == ref
        seq = seq[:pos1 - 1] + alt + seq[pos1 - 1 + len(ref):]
    return seq


def update(tmp_path, prgs, loci):
    from drprg_amd import Context
    f = str(tmp_path / "dr.prg")
    with open(f, "w") as o:
        for i, p in enumerate(prgs):
            o.write(">g%d\n%s\n" % (i, p))
    ctx = Context(f, 5, 7, device=-1, from_files=False)
    paths = str(tmp_path / "denovo_paths.txt")
    write_paths(paths, loci)
    out = str(tmp_path / "updated.dr.prg")
    n = ctx.update_prg_from_paths(paths, out)
    new = [l.rstrip("\n") for l in open(out) if not l.startswith(">")]
    # the updated file opens (parser, k-mer graph, index of this build) as well
    Context(out, 5, 7, device=-1, from_files=False)
    return n, new


PRG = "ACGTACGTTGCA 5 G 6 T 5 CCATGGATCCAT 7 AC 8  7 GGTTAACCGGTA 9 TT 11 A 12 C 11 GG 10 CA 9 TTGACCAGTCAA"
# consensus along the first alleles: ACGTACGTTGCA G CCATGGATCCAT AC GGTTAACCGGTA TT A GG TTGACCAGTCAA
FIRST = "ACGTACGTTGCA" + "G" + "CCATGGATCCAT" + "AC" + "GGTTAACCGGTA" + "TT" + "A" + "GG" + "TTGACCAGTCAA"


def test_the_walk_helper_spells_the_first_alleles():
    nodes = path_nodes(PRG, lambda s: 0)
    assert "".join(n[2] for n in nodes) == FIRST and all(PRG[a:b] == s for a, b, s in nodes)
    assert "".join(n[2] for n in path_nodes(PRG, lambda s: 1)) == "ACGTACGTTGCA" + "T" + "CCATGGATCCAT" + "" + "GGTTAACCGGTA" + "CA" + "TTGACCAGTCAA"
    assert FIRST in prg_language(PRG) and len(prg_language(PRG)) == 2 * 2 * 3


# 1-based positions on FIRST: 1-12 plain | 13 site 5 | 14-25 plain | 26-27 the AC / nothing site | 28-39 plain | 40-41 outer allele, left part |
# 42 the nested site | 43-44 outer allele, right part | 45-56 plain.  A case: [(pos1, bases of FIRST replaced, what replaces them)]
CASES = {
    "inside a node": [(5, 1, "G")],
    "first base of the locus": [(1, 1, "T")],
    "last base of the locus": [(56, 1, "C")],
    "last base before a site": [(12, 1, "C")],
    "first base behind a site": [(14, 1, "G")],
    "the site's own base": [(13, 1, "A")],
    "into a site from the left": [(12, 2, "CC")],
    "out of a site to the right": [(13, 2, "AA")],
    "across a site": [(12, 3, "TTT")],
    "deletion of a whole site and more": [(11, 5, "C")],
    "insertion in front of a site": [(13, 0, "TTT")],
    "insertion behind a site": [(14, 0, "GGG")],
    "insertion at the very start": [(1, 0, "GG")],
    "insertion at the very end": [(57, 0, "AC")],
    "inside the allele of the deletion site": [(27, 1, "T")],
    "across the deletion site": [(25, 4, "AAAA")],
    "inside the outer allele, before the nested site": [(41, 1, "G")],
    "the nested site's base": [(42, 1, "T")],
    "out of the nested site into the outer allele": [(42, 2, "CC")],
    "from the outer allele across the nested site": [(41, 3, "CCC")],
    "across the whole nested structure": [(39, 7, "GG")],
    "from plain sequence into the nested site": [(38, 5, "C")],
    "two variants in one node": [(3, 1, "C"), (8, 1, "A")],
    "two variants whose stretches are one site": [(12, 2, "CC"), (14, 1, "A")],
    "a variant in every part": [(5, 1, "G"), (13, 1, "A"), (20, 1, ""), (27, 1, "T"), (42, 2, "CC"), (55, 1, "T")],
}
CASES = {k: [(p, FIRST[p - 1:p - 1 + n], a) for p, n, a in v] for k, v in CASES.items()}


@pytest.mark.parametrize("name", sorted(CASES))
def test_update_places_the_variant_and_keeps_the_language(tmp_path, name):
    variants = CASES[name]
    old = prg_language(PRG)
    sample = apply(FIRST, variants)
    assert sample not in old, "the case must be novel"
    n, (new,) = update(tmp_path, [PRG], [("g0", path_nodes(PRG, lambda s: 0), variants)])
    assert n == len(variants)
    lang = prg_language(new)  # (raises if pandora's parser would refuse the string)
    assert old <= lang and sample in lang
    # nothing else than the variants' alleles has been added: every new sequence is an old one with some of the variants' stretches swapped in,
    # so it cannot be longer or shorter than the old ones by more than the variants change
    lens = {len(s) for s in old}
    slack = sum(abs(len(r) - len(a)) for _, r, a in variants)
    assert all(min(lens) - slack <= len(s) <= max(lens) + slack for s in lang)


def test_other_paths_and_untouched_loci(tmp_path):
    """the called path runs through second alleles (the deletion allele, the nested site's second allele); a second locus stays as it is"""
    other = "TTGACA 5 A 6 C 5 GGATCA"
    for choose, variants in ((lambda s: 1, [(30, "G", "C")]), (lambda s: [0, 1, 0, 1][s], [(25, "TG", "AA"), (43, "C", "G")])):
        nodes = path_nodes(PRG, choose)
        cons = "".join(n[2] for n in nodes)
        variants = [(p, cons[p - 1:p - 1 + len(r)], a) for p, r, a in variants]
        n, new = update(tmp_path, [PRG, other], [("g0", nodes, variants)])
        assert n == len(variants) and new[1] == other
        lang = prg_language(new[0])
        assert prg_language(PRG) <= lang and apply(cons, variants) in lang


def test_files_that_do_not_belong_to_the_prg_are_refused(tmp_path):
    from drprg_amd import Context
    from drprg_amd.pandora import DependencyError as DrprgError
    nodes = path_nodes(PRG, lambda s: 0)
    f = str(tmp_path / "dr.prg")
    open(f, "w").write(">g0\n%s\n" % PRG)
    ctx = Context(f, 5, 7, device=-1, from_files=False)
    for loci, what in (([("nope", nodes, [(5, "A", "G")])], "locus"), ([("g0", [(a + 1, b + 1, s) for a, b, s in nodes], [(5, "A", "G")])], "intervals"),
                       ([("g0", nodes, [(5, "C", "G")])], "does not lie")):
        write_paths(str(tmp_path / "p.txt"), loci)
        with pytest.raises(DrprgError) as e:
            ctx.update_prg_from_paths(str(tmp_path / "p.txt"), str(tmp_path / "o.prg"))
        assert what in str(e.value)
    open(str(tmp_path / "p.txt"), "w").write("garbage\n")
    with pytest.raises(DrprgError):
        ctx.update_prg_from_paths(str(tmp_path / "p.txt"), str(tmp_path / "o.prg"))
    # the reference's own example names no locus: nothing to do, the PRG comes back unchanged
    open(str(tmp_path / "p.txt"), "w").write("1 samples\nSample s\n0 loci with denovo variants\n")
    assert ctx.update_prg_from_paths(str(tmp_path / "p.txt"), str(tmp_path / "o.prg")) == 0
    assert open(str(tmp_path / "o.prg")).read() == ">g0\n%s\n" % PRG
