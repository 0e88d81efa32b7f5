"""Local assembly of the candidate regions (round 4; drprg_amd/csrc/denovo.cpp assemble_region): the de Bruijn half of discover, beside the
pile-up.  The product's assembly alone (DRPRG_HIP_DENOVO=dbg) against the oracle's own statement of the same rules
(oracle/oracle_denovo.py local_assembly), reads too short to hold both anchors of a region (the pile-up finds nothing, the assembly does),
and mixed samples: two novel alleles in one region, both reported, both placed in the updated PRG as alternative alleles of one new site,
the major one called.  Reference call sites: /root/reference/src/lib.rs:513-578 (pandora discover), /root/reference/src/predict.rs:247-284."""
import os
import re

import numpy as np
import pytest

from util import cluster_fraction, map_params, oracle_local_assembly, prg_spells, prg_walks

W, K = 11, 15


def _flip(s, i, n=0):
    return s[:i] + "ACGT".replace(s[i], "")[n] + s[i + 1:]


def _sample(tmp_path, oracle, haplotypes, read_len=150, n_reads=900, seed=4):
    """haplotypes: callable(ref of locus g1, position of the middle of its longest site-free stretch) -> [(share, sequence)] of locus g1;
    the other two loci follow their references"""
    from drprg_amd import Context, synth
    panel = synth.small_panel(seed=31, n_loci=3, length=900, site_every=60)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    rng = np.random.default_rng(seed)
    cur, best = 0, (0, 0)
    for seg in panel.trees[1]:
        if isinstance(seg, str):
            mid = cur + len(seg) // 2
            if len(seg) > best[0] and 250 <= mid <= len(panel.refs[1]) - 250:  # (well inside the locus: a region at its end has no room for an anchor)
                best = (len(seg), mid)
            cur += len(seg)
        else:
            cur += len(seg.alleles[0][0]) if isinstance(seg.alleles[0][0], str) else 0
    haps = haplotypes(panel.refs[1], best[1])
    reads = []
    for locus in range(3):
        for share, hap in (haps if locus == 1 else [(1.0, panel.refs[locus])]):
            h = np.frombuffer((synth.random_seq(rng, 200) + hap + synth.random_seq(rng, 200)).encode(), np.uint8)
            for s in rng.integers(0, len(h) - read_len, size=int(round(n_reads * share))):
                r = h[s:s + read_len]
                reads.append(synth._COMP[r[::-1]] if rng.random() < 0.5 else r)
    offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(read_len)
    bases = np.concatenate(reads)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, bases, offs)
    ctx = Context(prg, W, K, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=4000)
    md, er = map_params(K, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, W, K), W, K, md, cluster_fraction(er, K), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    ctx.set_threads(4)
    return ctx, panel, genes, fq, bases, offs, haps, best[1]


def _discover(ctx, fq, genes, out, mode=None):
    out.mkdir()
    old = os.environ.pop("DRPRG_HIP_DENOVO", None)
    if mode:
        os.environ["DRPRG_HIP_DENOVO"] = mode
    try:
        return ctx.discover_reads(fq, genes, str(out))
    finally:
        os.environ.pop("DRPRG_HIP_DENOVO", None)
        if old is not None:
            os.environ["DRPRG_HIP_DENOVO"] = old


def _apply(seq, pos0, ref, alt):
    assert seq[pos0:pos0 + len(ref)] == ref
    return seq[:pos0] + alt + seq[pos0 + len(ref):]


def _genotype_updated(tmp_path, oracle, ctx, panel, genes, bases, offs):
    from drprg_amd import Context
    new_prg = str(tmp_path / "updated.dr.prg")
    applied = ctx.update_prg(new_prg)
    prgs = [l.rstrip("\n") for l in open(new_prg) if not l.startswith(">")]
    ctx2 = Context(new_prg, W, K, device=-1, from_files=False)
    ctx2.set_opts(illumina=True, genome_size=4000)
    md, er = map_params(K, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(prgs, W, K), W, K, md, cluster_fraction(er, K), 10)
    ctx2.set_coverage(covg, prg_reads, int(offs[-1]))
    vcf = str(tmp_path / "updated.vcf")
    ctx2.genotype(genes, vcf)
    calls = []
    for line in open(vcf):
        if line.startswith("#"):
            continue
        t = line.rstrip("\n").split("\t")
        gt = t[9].split(":")[0]
        if gt not in ("0", "."):
            calls.append((t[0], int(t[1]), t[3], t[4].split(",")[int(gt) - 1]))
    return applied, prgs, calls


@pytest.mark.parametrize("kind", ["snp", "del", "ins"])
def test_the_assembly_alone_equals_the_oracle(tmp_path, oracle, kind):
    def haps(ref, pos):
        if kind == "snp":
            return [(1.0, _flip(ref, pos))]
        if kind == "del":
            return [(1.0, ref[:pos] + ref[pos + 3:])]
        return [(1.0, ref[:pos] + "ACGTA".replace(ref[pos], "")[:2] + "T" + ref[pos:])]

    ctx, panel, genes, fq, bases, offs, hs, pos = _sample(tmp_path, oracle, haps)
    got = _discover(ctx, fq, genes, tmp_path / "d", "dbg")
    want = oracle_local_assembly(tmp_path / "d", dict(zip(panel.names, panel.refs)), bases, offs)
    assert [(l, p - 1, r, a, s, n) for l, p, r, a, s, n in got] == sorted(want, key=lambda v: (v[0], v[1])) and len(got) >= 1
    locus, pos1, ref, alt, support, spanning = got[0]
    assert locus == "g1" and _apply(panel.refs[1], pos1 - 1, ref, alt) == hs[0][1] and support >= 20 and spanning >= support
    # ... and pile-up + assembly together report that one variant once (the pile-up's line: read counts)
    both = _discover(ctx, fq, genes, tmp_path / "d2")
    assert len(both) == 1 and _apply(panel.refs[1], both[0][1] - 1, both[0][2], both[0][3]) == hs[0][1]


def test_reads_too_short_to_hold_both_anchors(tmp_path, oracle):
    """100-base reads: a region is 1 + 2 x 22 bases of padding + 2 x 15 of anchors = 75 at least and up to 30 longer, the pile-up needs one read
    over all of it and mostly has none; the assembly joins reads that overlap"""
    ctx, panel, genes, fq, bases, offs, hs, pos = _sample(tmp_path, oracle, lambda ref, pos: [(1.0, ref[:pos] + "TCGGATACCGTTAGCAATGCCTAG" + ref[pos + 2:])],
                                                         read_len=100, n_reads=1400)
    pile = _discover(ctx, fq, genes, tmp_path / "p", "pileup")
    assert pile == []  # an insertion of 24 bases: no 100-base read holds both anchors
    got = _discover(ctx, fq, genes, tmp_path / "d")
    assert len(got) == 1 and got[0][0] == "g1" and _apply(panel.refs[1], got[0][1] - 1, got[0][2], got[0][3]) == hs[0][1]
    want = oracle_local_assembly(tmp_path / "d", dict(zip(panel.names, panel.refs)), bases, offs)
    assert [(l, p - 1, r, a, s, n) for l, p, r, a, s, n in got] == want
    applied, prgs, calls = _genotype_updated(tmp_path, oracle, ctx, panel, genes, bases, offs)
    assert applied == 1 and prg_spells(prgs[1], hs[0][1]) and not prg_spells(panel.prgs[1], hs[0][1])
    assert all(prg_spells(prgs[1], x) for x in prg_walks(panel.prgs[1], 30)) and prg_spells(prgs[1], panel.refs[1])
    assert len(calls) == 1 and calls[0][0] == "g1"


@pytest.mark.parametrize("layout", ["same position", "five bases apart"])
def test_a_mixed_sample_gives_two_alleles_of_one_new_site(tmp_path, oracle, layout):
    """65 % of the reads of g1 carry one novel SNP, 35 % another (at the same base / five bases further on): the pile-up reports the
    majority allele, the assembly adds the minor one; in the updated PRG they are two alleles of ONE new site (never one haplotype with
    both changes), and the major one is called"""
    def haps(ref, pos):
        second = pos if layout == "same position" else pos + 5
        return [(0.65, _flip(ref, pos, 0)), (0.35, _flip(ref, second, 1))]

    ctx, panel, genes, fq, bases, offs, hs, pos = _sample(tmp_path, oracle, haps, n_reads=1200)
    got = _discover(ctx, fq, genes, tmp_path / "d")
    spelled = [_apply(panel.refs[1], p - 1, r, a) for l, p, r, a, s, n in got if l == "g1"]
    assert sorted(spelled) == sorted(h for _, h in hs), got
    major = next(v for v in got if _apply(panel.refs[1], v[1] - 1, v[2], v[3]) == hs[0][1])
    minor = next(v for v in got if _apply(panel.refs[1], v[1] - 1, v[2], v[3]) == hs[1][1])
    assert major[4] >= 3 and minor[4] >= 3  # (the major one's line is the pile-up's: reads; the minor one's the assembly's: k-mer counts)
    # the assembly alone sees both as well, best supported first, exactly as the oracle does
    alone = _discover(ctx, fq, genes, tmp_path / "d2", "dbg")
    want = oracle_local_assembly(tmp_path / "d2", dict(zip(panel.names, panel.refs)), bases, offs)
    assert sorted((l, p - 1, r, a, s, n) for l, p, r, a, s, n in alone) == sorted(want) and len(alone) == 2
    by_support = sorted(alone, key=lambda v: -v[4])
    assert _apply(panel.refs[1], by_support[0][1] - 1, by_support[0][2], by_support[0][3]) == hs[0][1] and by_support[0][4] > by_support[1][4]
    _discover(ctx, fq, genes, tmp_path / "d3")  # (the update takes the last discover's variants: both modes)
    applied, prgs, calls = _genotype_updated(tmp_path, oracle, ctx, panel, genes, bases, offs)
    assert applied == 2
    assert prg_spells(prgs[1], hs[0][1]) and prg_spells(prgs[1], hs[1][1]) and not prg_spells(panel.prgs[1], hs[0][1])
    assert all(prg_spells(prgs[1], x) for x in prg_walks(panel.prgs[1], 30)) and prg_spells(prgs[1], panel.refs[1])
    assert len(re.findall(r" \d+ ", prgs[1])) == len(re.findall(r" \d+ ", panel.prgs[1])) + 4  # one site, three alleles: open, two separators, close
    if layout == "five bases apart":  # the haplotype with both changes is in neither read set and not in the PRG
        assert not prg_spells(prgs[1], _flip(hs[0][1], pos + 5, 1))
    assert [c for c in calls if c[0] == "g1"] and all(_apply(panel.refs[1], p - 1, r, a) == hs[0][1] for c, p, r, a in calls if c == "g1")


def test_discover_reads_a_multi_line_fastq_through_the_serial_reader(tmp_path, oracle):
    """a FASTQ whose sequence and quality run over several lines is not for the parallel ingest (it hands the file to the serial reader);
    discover's file pass does the same and finds what it finds in the one-line file"""
    ctx, panel, genes, fq, bases, offs, hs, pos = _sample(tmp_path, oracle, lambda ref, pos: [(1.0, _flip(ref, pos))])
    want = _discover(ctx, fq, genes, tmp_path / "one_line")
    assert len(want) == 1
    multi = str(tmp_path / "multi.fq")
    with open(fq) as src, open(multi, "w") as dst:
        lines = src.read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            h, s, p, q = lines[i:i + 4]
            dst.write("%s\n%s\n%s\n%s\n%s\n%s\n" % (h, s[:70], s[70:], p, q[:70], q[70:]))
    assert _discover(ctx, multi, genes, tmp_path / "multi_line") == want
