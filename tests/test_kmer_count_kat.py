"""The reference's own pin for the sketch / index semantics (rows a-4, a-5 of SURVEY.md section 8, and the allele -> k-mer rule of a-9).

No reference test feeds reads to pandora -- but every record of the seven pandora VCFs under /root/reference/tests/cases/predict/
prints, per allele, SUM_*_COVG and MEAN_*_COVG = floor(SUM / n) and GAPS = j / n: it leaks n, the NUMBER OF MINIMIZER K-MERS pandora took
the allele's statistics over.  n is a function of genes.fa (in the tree), the k-mer hash, the canonical rule, (w, k), the window rule,
the PRG sketch (which k-mers are k-mer-graph nodes) and the rule that assigns k-mer nodes to an allele.  tests/golden/kmer_count_kat.tsv
(tools/make_golden.py) holds, per allele, the set of n its statistics admit.  Here the oracle (oracle_index.c + oracle_vcf.c) and the
product index a PRG made of each fixture's own sites (genes.fa + that VCF's REF / ALT records) and their n is held against that set:

  * with minimap's hash64, k = 15 and the window size the fixture was made with, 318 of the 324 informative alleles agree (round 5: the
    MED_* fields joined SUM / MEAN / GAPS in the feasibility filter and made two more alleles informative; both agree) (3 of the 6
    that do not are the records SURVEY.md section 8a lists as hand-assembled); four control hashes reach a quarter to a third;
  * the fixtures come from two index generations: in*.vcf (fileDate 11/2022) agree at w = 14 (pandora's default), the three 2023 files
    at w = 11 (the w of tests/cases/predict/.config.toml, mtb-20230308) -- a scan over w peaks exactly there, a scan over k at 15;
  * they REJECT the allele -> k-mer rule this build had in rounds 1-3 (strict overlap with the bare allele: 204 of 262 on in.vcf) and
    pin the one it has now: the range is the allele as printed (padding base included) and the k-mer that ends exactly where it starts
    counts (259 of 262);
  * they confirm pandora's forward-greedy PRG sketch: eight alleles hold a k-mer node that is a window minimizer on no walk of the PRG,
    and all eight counts need that node.
"""
import os
from collections import Counter

import pytest

from util import GOLDEN, flat_prgs_from_sites, product_sites, read_fasta_dict

K = 15
# the window size each fixture's index was built with (asserted by test_window_and_kmer_size_scan, not assumed)
FILE_W = {"in.vcf": 14, "in2.vcf": 14, "in3.vcf": 14, "in4.vcf": 14, "ERR4796933.pandora.vcf": 11, "SRR6824468.vcf": 11,
          "ERR2510634.drprg.vcf": 11}
# every allele whose count disagrees, classified (file, chrom, pos, allele): why
KNOWN_MISMATCHES = {
    ("in.vcf", "ddn", 627, 1): "hand-edited record (SURVEY 8a: GAPS edited, no n fits SUM / MEAN / GAPS)",
    ("in.vcf", "katG", 1044, 0): "hand-edited record (SURVEY 8a: MEAN edited, SUM is 0)",
    ("in4.vcf", "fabG1", 92, 1): "hand-assembled record (SURVEY 8a: spliced from another sample; VC=SNP on a two-base REF)",
    # (round 5, test_the_two_deletions below: no rule for padded records does better than the default one, every uniform shift of the rule
    # loses 40+ alleles, and all three records lie within w + k - 1 bases of catalogue sites the fixture does not list)
    ("in.vcf", "embA", 69, 1): "unseen neighbours (panel.bcf embA 85, 89): deletion GC -> G inside a run of C, oracle 3 (one of them covered by REF "
                               "reads: it spells a REF k-mer), fixture 4 (one covered, three not): one uncovered k-mer node short",
    ("in.vcf", "gid", 160, 0): "unseen neighbours (panel.bcf gid 155-157, 176-178): deletion GC -> G, oracle 4, fixture 3: the k-mer that ends on the "
                               "padding base is a node here (a minimizer on the deletion's walk only) and is not among pandora's",
    ("in.vcf", "gyrA", 362, 0): "unseen neighbours: codon 88-91 of gyrA holds seven catalogue sites (panel.bcf 362-371) that in.vcf does not list; oracle 3, fixture 4",
}


def _kat():
    rows = {}
    for line in open(os.path.join(GOLDEN, "kmer_count_kat.tsv")):
        if line.startswith("#"):
            continue
        f, chrom, pos, ref, alts, a, sf, sr, mf, mr, gaps, feas, excl = line.rstrip("\n").split("\t")
        rec = rows.setdefault(f, {}).setdefault((chrom, int(pos), ref, alts), dict(chrom=chrom, pos=int(pos), ref=ref, alts=alts.split(","), feasible=[]))
        rec["feasible"].append(None if feas == "*" else ([] if feas == "-" else [int(x) for x in feas.split(",")]))
    return {f: list(v.values()) for f, v in rows.items()}


KAT = _kat()
GENES = read_fasta_dict(os.path.join(GOLDEN, "downstream", "genes.fa"))


def _oracle_counts(oracle, fname, w, k=K):
    """{(chrom, pos, allele): (n of the oracle, feasible set)} over the informative alleles of one fixture"""
    recs = KAT[fname]
    names, prgs, placed = flat_prgs_from_sites(GENES, recs)
    pred = {}
    for n, s in zip(names, prgs):
        if any(r["chrom"] == n for r in recs):
            for r in oracle.vcf_sites(s, w, k, GENES[n])[0]:
                pred[(n, r["pos"], r["ref"])] = r
    out = {}
    for r in recs:
        pr = pred.get((r["chrom"], r["pos"], r["ref"]))
        if pr is None:
            continue
        for a, feas in enumerate(r["feasible"]):
            if feas is None:
                continue
            if a and r["alts"][a - 1] not in pr["alts"]:
                continue
            ai = 0 if a == 0 else 1 + pr["alts"].index(r["alts"][a - 1])  # (pandora lists ALTs in PRG order, this build in byte order)
            out[(r["chrom"], r["pos"], a)] = (len(pr["knodes"][ai]), feas)
    return out


def _score(counts):
    return sum(n in feas for n, feas in counts.values()), len(counts)


def test_golden_file_shape():
    assert set(KAT) == set(FILE_W)
    n_inf = sum(f is not None and f != [] for recs in KAT.values() for r in recs for f in r["feasible"])
    assert n_inf == 323  # (321 from SUM / MEAN / GAPS; the medians make two more alleles informative: round 5)
    # the integer-mean rule itself: ahpC:19 allele 2 of in.vcf (SUM 29 over 3 k-mers -> MEAN 9) admits exactly n = 3
    r = next(r for r in KAT["in.vcf"] if (r["chrom"], r["pos"]) == ("ahpC", 19))
    assert r["feasible"][0] == [5] and r["feasible"][2] == [3]


def test_hash64_explains_the_fixture_counts_and_control_hashes_do_not(oracle):
    rates = {}
    try:
        for mode, label in ((0, "hash64"), (1, "identity"), (2, "fibonacci"), (3, "splitmix64"), (4, "fmix64")):
            oracle.set_hash_mode(mode, 12345)
            ok = tot = 0
            per = {}
            for f, w in FILE_W.items():
                c = _oracle_counts(oracle, f, w)
                per[f] = _score(c)
                ok += per[f][0]
                tot += per[f][1]
            rates[label] = (ok, tot, per)
    finally:
        oracle.set_hash_mode(0)
    ok, tot, per = rates["hash64"]
    assert (ok, tot) == (318, 324), rates["hash64"]
    assert per["in.vcf"] == (261, 266) and per["SRR6824468.vcf"] == (16, 16) and per["ERR2510634.drprg.vcf"] == (9, 9) and per["in2.vcf"] == (13, 13)
    best_control = max(v[0] for k, v in rates.items() if k != "hash64")
    assert ok >= 2 * best_control and best_control <= 0.4 * tot, {k: v[:2] for k, v in rates.items()}


def test_every_mismatch_is_accounted_for(oracle):
    bad = {}
    for f, w in FILE_W.items():
        for (chrom, pos, a), (n, feas) in _oracle_counts(oracle, f, w).items():
            if n not in feas:
                bad[(f, chrom, pos, a)] = (n, feas[:4])
    assert set(bad) == set(KNOWN_MISMATCHES), bad


def test_window_and_kmer_size_scan(oracle):
    """the count agreement as a function of w peaks at the w the index of each fixture generation was built with, and at k = 15"""
    for f in ("in.vcf", "SRR6824468.vcf"):
        by_w = {w: _score(_oracle_counts(oracle, f, w))[0] for w in (9, 10, 11, 12, 13, 14, 15, 16, 19)}
        assert max(by_w, key=by_w.get) == FILE_W[f], by_w
        others = max(v for w, v in by_w.items() if w != FILE_W[f])
        assert by_w[FILE_W[f]] >= others + (30 if f == "in.vcf" else 2), by_w
    by_k = {k: _score(_oracle_counts(oracle, "in.vcf", 14, k))[0] for k in (13, 14, 15, 16, 17)}
    assert max(by_k, key=by_k.get) == 15 and by_k[15] >= 2 * max(v for k, v in by_k.items() if k != 15), by_k
    # the small files of each generation agree with their own w and not with the other one
    for f, w in FILE_W.items():
        other = 25 - w
        assert _score(_oracle_counts(oracle, f, w))[0] > _score(_oracle_counts(oracle, f, other))[0] or f in ("in4.vcf",), f


def test_fixtures_reject_the_strict_overlap_rule(oracle):
    try:
        oracle.lib.orc_vcf_set_overlap_rule(1)
        old = _score(_oracle_counts(oracle, "in.vcf", 14))
    finally:
        oracle.lib.orc_vcf_set_overlap_rule(0)
    new = _score(_oracle_counts(oracle, "in.vcf", 14))
    assert old == (206, 266) and new == (261, 266)
    # every allele the old rule got wrong and the new one gets right has one k-mer MORE in the fixture than the old rule counted
    oracle.lib.orc_vcf_set_overlap_rule(1)
    try:
        c_old = _oracle_counts(oracle, "in.vcf", 14)
    finally:
        oracle.lib.orc_vcf_set_overlap_rule(0)
    diffs = Counter(min(feas) - n for n, feas in c_old.values() if feas and n not in feas)
    assert set(diffs) == {1}, diffs


def test_fixtures_confirm_the_forward_greedy_nodes(oracle):
    """pandora continues a k-mer node along every walk, also along walks on which that node is no window minimizer:
    alleles that hold such a node have a count that needs it"""
    recs = KAT["in.vcf"]
    names, prgs, placed = flat_prgs_from_sites(GENES, recs)
    with_node = need_node = 0
    for n, s in zip(names, prgs):
        conf = oracle.sketch_prg(s, 14, K, walk_check=True)["walk"]["mask"]
        pred = {(r["pos"], r["ref"]): r for r in oracle.vcf_sites(s, 14, K, GENES[n])[0]}
        for r in recs:
            pr = pred.get((r["pos"], r["ref"])) if r["chrom"] == n else None
            if pr is None:
                continue
            for a, feas in enumerate(r["feasible"]):
                if not feas or (a and r["alts"][a - 1] not in pr["alts"]):
                    continue
                ids = pr["knodes"][0 if a == 0 else 1 + pr["alts"].index(r["alts"][a - 1])]
                extra = [i for i in ids if not conf[i - 1]]
                if extra:
                    with_node += 1
                    need_node += (len(ids) in feas) and (len(ids) - len(extra) not in feas)
    assert with_node == 8 and need_node == 8


def test_product_counts_equal_oracle_counts(tmp_path, oracle):
    """the product's index + genotyper on the same PRGs: the same n (the whole node sets: tests/test_vcf_sites.py), hence the same agreement"""
    for f in ("in.vcf", "SRR6824468.vcf", "ERR2510634.drprg.vcf"):
        w = FILE_W[f]
        recs = KAT[f]
        names, prgs, placed = flat_prgs_from_sites(GENES, recs)
        keep = [i for i, n in enumerate(names) if any(r["chrom"] == n for r in recs)]
        names, prgs = [names[i] for i in keep], [prgs[i] for i in keep]
        d = tmp_path / f
        d.mkdir()
        prod = product_sites(names, prgs, [GENES[n] for n in names], w, K, d)
        want = _oracle_counts(oracle, f, w)
        got_ok = 0
        for r in recs:
            pr = next((x for x in prod[r["chrom"]] if (x["pos"], x["ref"]) == (r["pos"], r["ref"])), None) if r["chrom"] in prod else None
            if pr is None:
                continue
            for a, feas in enumerate(r["feasible"]):
                if feas is None or (a and r["alts"][a - 1] not in pr["alts"]):
                    continue
                n = len(pr["knodes"][0 if a == 0 else 1 + pr["alts"].index(r["alts"][a - 1])])
                assert n == want[(r["chrom"], r["pos"], a)][0]
                got_ok += n in feas
        assert got_ok == _score(want)[0]


def _padded(r):
    alls = [r["ref"]] + r["alts"]
    return len({len(a) for a in alls}) > 1 and all(a[:1] == r["ref"][:1] for a in alls) and min(len(a) for a in alls) == 1


def _score_all(oracle):
    ok = tot = 0
    for f, w in FILE_W.items():
        a, b = _score(_oracle_counts(oracle, f, w))
        ok += a
        tot += b
    return ok, tot


def test_the_two_deletions(oracle):
    """in.vcf embA:69 ALT (oracle 3, fixture 4) and gid:160 REF (oracle 4, fixture 3), both GC -> G.  What the fixtures
    say about them (tools/indel_rule_scan.py prints the whole table, DESIGN.md section 5 quotes it):
      * the allele -> k-mer rule is a sharp optimum: moving either end of the range by one base for EVERY record loses 42 alleles or more,
        and no rule for the records with a padding base -- printed or bare range, either end moved by -1 .. +2, an empty allele made one
        base wide -- explains more than the default's 15 of their 18 informative alleles; none explains the two deletions together;
      * embA:69: the deleted C lies in a run (GCCCT): the ALT k-mer that crosses the deletion by one base spells the REF k-mer at the same
        place (same hash), so REF reads cover it -- the fixture's ALT has exactly one covered k-mer (35 / 36 against the REF k-mers' 40)
        and three bare ones; this build has the covered one and two bare ones: ONE uncovered node short;
      * all three genuine mismatches (with gyrA:362) lie within w + k - 1 bases of a catalogue site (panel.bcf) that in.vcf does not list,
        and the three mismatches that do not are the hand-edited records: the counts there depend on PRG structure the fixture does not show
        (126 of the 129 alleles with such a neighbour agree all the same)."""
    L = oracle.lib
    base = _score_all(oracle)
    assert base == (318, 324)
    try:
        L.orc_vcf_set_overlap_rule(3)
        for dl, dr in ((-1, 0), (1, 0), (0, -1), (0, 1)):
            L.orc_vcf_set_all_rule(dl, dr)
            assert _score_all(oracle)[0] <= 276, (dl, dr)
        L.orc_vcf_set_overlap_rule(2)
        best = 0
        for bare in (0, 1):
            for dl in (-1, 0, 1, 2):
                for dr in (-1, 0, 1):
                    for ext in (0, 1):
                        L.orc_vcf_set_padded_rule(bare, dl, dr, ext)
                        c = _oracle_counts(oracle, "in.vcf", 14)
                        both = c[("embA", 69, 1)][0] in c[("embA", 69, 1)][1] and c[("gid", 160, 0)][0] in c[("gid", 160, 0)][1]
                        assert not both, (bare, dl, dr, ext)
                        best = max(best, _score_all(oracle)[0])
        assert best == 318
    finally:
        L.orc_vcf_set_overlap_rule(0)
        L.orc_vcf_set_padded_rule(0, 0, 0, 0)
        L.orc_vcf_set_all_rule(0, 0)
    # embA:69: the ALT k-mer that crosses the deletion by one base and the REF k-mer over the deleted base share their hash
    names, prgs, _ = flat_prgs_from_sites(GENES, KAT["in.vcf"])
    prg = dict(zip(names, prgs))["embA"]
    sk = oracle.sketch_prg(prg, 14, K, paths=True)
    rec = next(r for r in oracle.vcf_sites(prg, 14, K, GENES["embA"])[0] if r["pos"] == 69)
    ref_h = {int(sk["hash"][i - 1]) for i in rec["knodes"][0]}
    alt_h = [int(sk["hash"][i - 1]) for i in rec["knodes"][1]]
    assert len(alt_h) == 3 and sum(h in ref_h for h in alt_h) == 1
    # every mismatch that is not a hand-edited record has a catalogue site within w + k - 1 bases that the fixture does not list
    from drprg_amd import bcf_lite
    sites = {}
    for r in bcf_lite.read_bcf(os.path.join(GOLDEN, "downstream", "panel.bcf")):
        sites.setdefault(r["chrom"], set()).add(r["pos"] + 1)  # (bcf_lite positions are 0-based)
    near = {True: [0, 0], False: [0, 0]}
    for f, w in FILE_W.items():
        listed = {}
        for r in KAT[f]:
            listed.setdefault(r["chrom"], set()).update(range(r["pos"], r["pos"] + len(r["ref"])))
        for (chrom, pos, a), (n, feas) in _oracle_counts(oracle, f, w).items():
            unseen = any(abs(p - pos) <= w + K - 1 and not any(q in listed.get(chrom, ()) for q in range(p, p + 3)) for p in sites.get(chrom, ()))
            near[unseen][0] += n in feas
            near[unseen][1] += 1
            if n not in feas:
                assert unseen == ("hand" not in KNOWN_MISMATCHES[(f, chrom, pos, a)]), (f, chrom, pos, a)
    assert near == {True: [126, 129], False: [192, 195]}
