"""`drprg predict` end to end on the GPU box: reads -> hot path -> pandora_genotyped.vcf -> annotated VCF -> JSON,
on an mtb-like index built from the reference's test index files (tests/golden/downstream: genes.fa, panel.bcf,
rules.csv, .config.toml) with the panel variants as PRG sites."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from util import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
DS = os.path.join(GOLDEN, "downstream")
BIN = os.path.join(ROOT, "drprg_amd", "bin")


def _make_index(tmp_path):
    from drprg_amd import synth
    idx = tmp_path / "idx"
    idx.mkdir()
    for f in (".config.toml", "genes.fa", "genes.fa.fai", "panel.bcf", "panel.bcf.csi", "rules.csv"):
        shutil.copy(os.path.join(DS, f), idx / f)
    (idx / "msas").mkdir()
    panel, sites = synth.panel_from_index_dir(str(idx))
    panel.write(str(idx / "dr.prg"))
    r = subprocess.run([os.path.join(BIN, "pandora"), "index", "-t", "4", "-w", "11", "-k", "15", str(idx / "dr.prg")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return idx, panel, sites


def _reads(panel, choose, n_reads, seed):
    from drprg_amd import synth
    rng = np.random.default_rng(seed)
    spacer = synth.random_seq(rng, 300)
    genome = spacer + spacer.join(synth.haplotype_with(t, lambda i, g=gi: choose(g, i)) for gi, t in enumerate(panel.trees)) + spacer
    g = np.frombuffer(genome.encode(), np.uint8)
    starts = rng.integers(0, len(g) - 150, size=n_reads)
    block = g[starts[:, None] + np.arange(150)]
    rev = rng.random(n_reads) < 0.5
    block[rev] = synth._COMP[block[rev][:, ::-1]]
    return block.reshape(-1), np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(150)


def _predict(tmp_path, idx, bases, offs, name, illumina=True, env=None, outname=None):
    from drprg_amd import synth
    fq = str(tmp_path / f"{name}.fq")
    if not os.path.exists(fq):
        synth.write_fastq(fq, bases, offs)
    out = tmp_path / f"out_{outname or name}"
    r = subprocess.run([os.path.join(BIN, "drprg"), "predict", "-x", str(idx), "-i", fq, "-o", str(out), "-s", name, "-v"]
                       + (["-I"] if illumina else []), capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    _predict.stderr = r.stderr
    for f in ("pandora_genotyped.vcf", f"{name}.drprg.vcf", f"{name}.drprg.json", "discover/denovo_paths.txt"):
        assert (out / f).exists(), f
    return json.load(open(out / f"{name}.drprg.json")), out


def test_reference_sample_is_susceptible_and_resistant_sample_is_called(tmp_path):
    idx, panel, sites = _make_index(tmp_path)
    gene_ix = {g: i for i, g in enumerate(panel.names)}
    # wild type: every site takes the reference allele -> every drug S, no evidence, all 18 genes present
    bases, offs = _reads(panel, lambda g, i: 0, 14000, seed=1)
    wt, _ = _predict(tmp_path, idx, bases, offs, "wt")
    assert wt["sample"] == "wt" and wt["genes"]["absent"] == [] and len(wt["genes"]["present"]) == 18
    assert all(v["predict"] == "S" and v["evidence"] == [] for v in wt["susceptibility"].values()), wt["susceptibility"]
    assert wt["version"]["index"] == "mtb-20230308"
    # mutant: one panel variant in rrs and one in katG take their first alternate allele
    want = {}
    for gene in ("rrs", "katG", "gyrA"):
        k = len(sites[gene]) // 2
        want[gene] = (k, sites[gene][k][0])
    bases, offs = _reads(panel, lambda g, i: 1 if any(gene_ix[gn] == g and i == k for gn, (k, _) in want.items()) else 0,
                         14000, seed=2)
    mut, out = _predict(tmp_path, idx, bases, offs, "mut")
    evid = {(e["gene"], e["variant"]) for v in mut["susceptibility"].values() if v["predict"] == "R" for e in v["evidence"]}
    for gene, (k, vid) in want.items():
        assert any(g == gene for g, _ in evid), (gene, vid, evid)
    resistant = {d for d, v in mut["susceptibility"].items() if v["predict"] == "R"}
    assert resistant and resistant != set(mut["susceptibility"])
    # the genotyped VCF calls the alternate allele at exactly the mutated sites
    alt_calls = set()
    for line in open(out / "pandora_genotyped.vcf"):
        if line.startswith("#"):
            continue
        t = line.split("\t")
        if t[9].split(":")[0] not in ("0", "."):
            alt_calls.add((t[0], int(t[1])))
    expect_calls = {(gene, sites[gene][k][1] + 1) for gene, (k, _) in want.items()}
    assert expect_calls <= alt_calls and len(alt_calls) <= len(expect_calls) + 2, (alt_calls, expect_calls)
    # the report surface of the reference: <sample>.drprg.bcf (/root/reference/src/predict.rs:429-431) beside the text VCF,
    # same records; discover's outputs (/root/reference/src/predict.rs:247-256) with the loud warning on stderr
    from bcf_decode import decode
    _, recs = decode(str(out / "mut.drprg.bcf"))
    text = [l.split("\t") for l in open(out / "mut.drprg.vcf") if not l.startswith("#")]
    assert [(r["chrom"], r["pos"] + 1, r["alleles"][0]) for r in recs] == [(t[0], int(t[1]), t[3]) for t in text] and len(recs) > 100
    assert [dict(r["info"]).get("PREDICT") for r in recs] == [dict(kv.split("=", 1) for kv in t[7].split(";") if "=" in kv).get("PREDICT") for t in text]
    assert (out / "discover" / "denovo_paths.txt").exists() and (out / "discover" / "candidate_regions.tsv").exists()


def test_absent_gene_is_reported(tmp_path):
    idx, panel, sites = _make_index(tmp_path)
    pnca = panel.names.index("pncA")
    from drprg_amd import synth
    # a sample without pncA: gene absence -> Pyrazinamide R with the gene_absent evidence (rules.csv: absence,pncA,,,Pyrazinamide)
    trees = [t for i, t in enumerate(panel.trees) if i != pnca]
    sub = synth.Panel([n for i, n in enumerate(panel.names) if i != pnca], trees)
    bases, offs = _reads(sub, lambda g, i: 0, 14000, seed=3)
    res, _ = _predict(tmp_path, idx, bases, offs, "nopnca")
    assert res["genes"]["absent"] == ["pncA"]
    pza = res["susceptibility"]["Pyrazinamide"]
    assert pza["predict"] == "R" and pza["evidence"][0]["variant"] == "gene_absent" and pza["evidence"][0]["gene"] == "pncA"


@pytest.mark.parametrize("tech", ["illumina", "nanopore"])
def test_off_panel_variant_is_discovered_and_reported_as_unknown(tmp_path, tech):
    """The reference's reason for running discover (/root/reference/src/predict.rs:247-302): a non-synonymous variant that is
    NOT in the panel must come out as an unknown (`U`) call, not as susceptible.  The sample carries a missense SNP in the
    middle of katG's coding sequence at a position the panel holds no site for: discover finds it in the reads, the PRG in
    the output directory gains the site, the reads are mapped again, the VCF calls the new allele and the report holds the
    evidence with prediction U for katG's drug.  Both technologies: 150-base accurate reads with -I, and 2-kb reads with 5 %
    errors without (the pile-up then takes the column-wise majority of the aligned reads)."""
    from drprg_amd import synth
    idx, panel, sites = _make_index(tmp_path)
    g = panel.names.index("katG")
    ref = panel.refs[g]
    # a codon of katG (padding 100: the CDS starts at offset 100) whose neighbourhood holds no panel site
    taken = [p for _, p, r, _ in sites["katG"] for p in range(p - 40, p + len(r) + 40)]
    pos = next(p for p in range(100 + 3 * 250, len(ref) - 400, 3) if p not in taken and p + 1 not in taken and p + 2 not in taken)
    codon = ref[pos:pos + 3]
    # second base of the codon -> a change that is never synonymous for the standard table except ... pick one that alters the residue
    alt_base = next(b for b in "ACGT" if b != codon[1] and _aa(codon[0] + b + codon[2]) not in (_aa(codon), "*"))
    mutated = ref[:pos + 1] + alt_base + ref[pos + 2:]
    rng = np.random.default_rng(7)
    spacer = synth.random_seq(rng, 300)
    genome = spacer + spacer.join(mutated if gi == g else r for gi, r in enumerate(panel.refs)) + spacer
    gn = np.frombuffer(genome.encode(), np.uint8)
    if tech == "illumina":
        starts = rng.integers(0, len(gn) - 150, size=16000)
        block = gn[starts[:, None] + np.arange(150)]
        rev = rng.random(len(starts)) < 0.5
        block[rev] = synth._COMP[block[rev][:, ::-1]]
        bases, offs = block.reshape(-1), np.arange(len(starts) + 1, dtype=np.uint64) * np.uint64(150)
    else:
        from types import SimpleNamespace
        bases, offs = synth.sample_long_reads(SimpleNamespace(haps=[gn], lens=np.array([gn.size], dtype=np.int64)), 60 * gn.size // 2000,
                                              seed=21, mean_len=2000, min_len=500, max_len=gn.size - 1)
    res, out = _predict(tmp_path, idx, bases, offs, "novel", illumina=tech == "illumina")
    variants = [l.split("\t") for l in open(out / "discover" / "denovo_variants.tsv") if not l.startswith("#")]
    assert len(variants) == 1 and variants[0][0] == "katG" and int(variants[0][1]) == pos + 2 and variants[0][2:4] == [codon[1], alt_base]
    assert (out / "updated.dr.prg").exists() and (out / "updated.dr.prg.k15.w11.idx").exists()
    calls = [t.split("\t") for t in open(out / "pandora_genotyped.vcf") if t.startswith("katG\t")]
    alt_calls = [t for t in calls if t[9].split(":")[0] not in ("0", ".")]
    assert len(alt_calls) == 1 and int(alt_calls[0][1]) == pos + 2 and alt_calls[0][3] == codon[1] and alt_calls[0][4] == alt_base
    evid = [(d, e["variant"], v["predict"]) for d, v in res["susceptibility"].items() for e in v["evidence"] if e["gene"] == "katG"]
    aa_pos = (pos - 100) // 3 + 1
    assert evid and all(p == "U" for _, _, p in evid), res["susceptibility"]
    assert all(var == f"{_aa(codon)}{aa_pos}{_aa(codon[0] + alt_base + codon[2])}" for _, var, _ in evid), evid
    assert {d for d, _, _ in evid} == {"Isoniazid"}
    # every other drug stays susceptible
    assert all(v["predict"] == "S" for d, v in res["susceptibility"].items() if d != "Isoniazid")
    # the file was read ONCE: discover took its reads from the blocks the mapping pass left in HBM, and so did the second mapping
    # pass (include/drprg_hip.h: drprg_hip_keep_reads) -- and reading the file three times instead gives the same bytes
    assert "(reads resident in device memory)" in _predict.stderr and "reads mapped again (resident in device memory)" in _predict.stderr
    res2, out2 = _predict(tmp_path, idx, bases, offs, "novel", illumina=tech == "illumina", env={"DRPRG_HIP_KEEP_READS_GB": "0"}, outname="novel_file")
    assert "(reads from the file)" in _predict.stderr and "reads mapped again (from the file)" in _predict.stderr
    import re
    # (the record IDs of the annotated VCF are random, as the reference's are: Uuid::new_v4, /root/reference/src/predict.rs:447)
    strip = lambda t: re.sub(r'(?m)^(\S+\t\d+\t)[0-9a-f]{8}\t', r'\1ID\t', re.sub(r'"vcfid": "[0-9a-f]{8}"', '"vcfid": "ID"', t.replace("out_novel_file", "out_novel")))
    for f in ("pandora_genotyped.vcf", "novel.drprg.vcf", "novel.drprg.json", "discover/denovo_paths.txt", "discover/denovo_variants.tsv", "updated.dr.prg"):
        assert strip(open(out2 / f).read()) == strip(open(out / f).read()), f


_CODONS = {a + b + c: aa for (a, b, c), aa in zip(((x, y, z) for x in "TCAG" for y in "TCAG" for z in "TCAG"),
                                                  "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG")}


def _aa(codon):
    return _CODONS[codon]


def test_gzip_input_is_read_once_too(tmp_path):
    """the same off-panel sample as `.fq.gz` (what the reference's users feed it, docs/src/guide/predict.md): inflated once, discover and
    the second mapping pass take the reads from HBM, and the calls equal those of the run that inflates the file three times"""
    from drprg_amd import synth
    idx, panel, sites = _make_index(tmp_path)
    g = panel.names.index("katG")
    ref = panel.refs[g]
    taken = [p for _, p, r, _ in sites["katG"] for p in range(p - 40, p + len(r) + 40)]
    pos = next(p for p in range(100 + 3 * 250, len(ref) - 400, 3) if p not in taken and p + 1 not in taken and p + 2 not in taken)
    codon = ref[pos:pos + 3]
    alt_base = next(b for b in "ACGT" if b != codon[1] and _aa(codon[0] + b + codon[2]) not in (_aa(codon), "*"))
    mutated = ref[:pos + 1] + alt_base + ref[pos + 2:]
    rng = np.random.default_rng(11)
    spacer = synth.random_seq(rng, 300)
    gn = np.frombuffer((spacer + spacer.join(mutated if gi == g else r for gi, r in enumerate(panel.refs)) + spacer).encode(), np.uint8)
    starts = rng.integers(0, len(gn) - 150, size=16000)
    block = gn[starts[:, None] + np.arange(150)]
    rev = rng.random(len(starts)) < 0.5
    block[rev] = synth._COMP[block[rev][:, ::-1]]
    fq = str(tmp_path / "s.fq.gz")
    synth.write_fastq(fq, block.reshape(-1), np.arange(len(starts) + 1, dtype=np.uint64) * np.uint64(150), gz=True)
    outs = {}
    for name, env in (("hbm", {}), ("file", {"DRPRG_HIP_KEEP_READS_GB": "0"})):
        out = tmp_path / name
        r = subprocess.run([os.path.join(BIN, "drprg"), "predict", "-x", str(idx), "-i", fq, "-o", str(out), "-s", "s", "-I", "-v"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert ("reads mapped again (resident in device memory)" in r.stderr) == (name == "hbm"), r.stderr
        outs[name] = (open(out / "pandora_genotyped.vcf").read(), json.load(open(out / "s.drprg.json"))["susceptibility"]["Isoniazid"]["predict"],
                      open(out / "discover" / "denovo_variants.tsv").read())
    assert outs["hbm"] == outs["file"] and outs["hbm"][1] == "U"
