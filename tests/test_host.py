"""CPU-only tests: C ABI surface, PRG parsing, index build / round trip, genotyper, CLI, loud failure
without a GPU.  No compute call of the hot path is made here (there is no CPU fallback to call)."""
import os
import re
import subprocess

import numpy as np
import pytest

from util import GOLDEN, ROOT, cluster_fraction, map_params

PANDORA = os.path.join(ROOT, "drprg_amd", "bin", "pandora")


def test_library_exports_every_declared_symbol():
    import ctypes
    from drprg_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "drprg_hip.h")).read()
    declared = set(re.findall(r"\b(drprg_hip_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/drprg_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_product_does_not_reference_the_oracle():
    for d, _, fs in os.walk(os.path.join(ROOT, "drprg_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt.replace("oracle/oracle.c consumes", ""), f


def test_prg_parse_reference_fixture(tmp_path):
    """tests/golden/prg_syntax/dr.prg = /root/reference/tests/cases/expected/dr.prg; the node-interval
    convention is pinned by the denovo_paths.txt embedded in /root/reference/src/lib.rs:3010-3038."""
    from drprg_amd import Context
    prg = os.path.join(GOLDEN, "prg_syntax", "dr.prg")
    ctx = Context(prg, 11, 15, device=-1, from_files=False)
    assert ctx.n_prgs == 2
    starts, ends, n_sites = ctx.prg_nodes(0)  # gid
    want = {}
    gid = False
    for line in open(os.path.join(GOLDEN, "denovo_paths_example.txt")):
        if line.strip() == "gid":
            gid = True
        elif line.strip() == "ahpC":
            gid = False
        m = re.match(r"\((\d+) \[(\d+), (\d+)\) ([ACGT]*)\)", line)
        if m and gid:
            want[int(m.group(1))] = (int(m.group(2)), int(m.group(3)), m.group(4))
    assert len(want) == 8
    # gid has sites 5..33 (15 sites, one of them nested), pncA sites 5..21
    assert n_sites == 15
    assert ctx.prg_nodes(1)[2] == 9
    text = open(prg).read().splitlines()[1]
    assert (int(starts[0]), int(ends[0])) == (0, 116) and text[119:120] == "C" and (int(starts[1]), int(ends[1])) == (119, 120)
    # The embedded denovo_paths.txt comes from the real mtb index, whose gid PRG differs from the fixture above;
    # rebuild the PRG prefix it implies (unlisted alleles filled with one base) and check ids + intervals.
    s = want[0][2] + " 5 C 6 T 5 " + want[3][2] + " 7 C 8 T 7 " + want[6][2] + " 9 " + "A" * 24 + " 10 " + want[8][2] \
        + " 10 G 9 " + want[10][2] + " 11 GC 12 G 11 " + "ACGTACGTACGTACGTACGTACGTACGTACGT"
    p2 = tmp_path / "denovo.prg"
    p2.write_text(f">gid\n{s}\n")
    c2 = Context(str(p2), 11, 15, device=-1, from_files=False)
    st, en, _ = c2.prg_nodes(0)
    for node, (a, b, seq) in want.items():
        assert (int(st[node]), int(en[node])) == (a, b), node
        assert s[a:b] == seq


def test_prg_parse_rejects_malformed(tmp_path):
    from drprg_amd import Context, DependencyError
    for bad in ["ACGT 5 A 6 C", "ACGT 5 A 5 ", "AC 6 T 5 G", "ACXT", "ACGT 5 A 6 C 7 GG"]:
        p = tmp_path / "bad.prg"
        p.write_text(f">x\n{bad}\n")
        with pytest.raises(DependencyError):
            Context(str(p), 3, 5, device=-1, from_files=False)


def _gfa_nodes(path):
    nodes = {}
    for line in open(path):
        t = line.rstrip("\n").split("\t")
        if t[0] == "S":
            nodes[int(t[1])] = [(int(a), int(b)) for a, b in re.findall(r"\[(\d+), (\d+)\)", t[2])]
    return nodes


def test_index_files_and_roundtrip(tmp_path, oracle):
    from drprg_amd import Context, Pandora, synth
    panel = synth.small_panel(seed=5)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg)
    Pandora().index_with(prg, ["-t", "2", "-w", "11", "-k", "15"])
    assert os.path.exists(prg + ".k15.w11.idx") and os.path.isdir(tmp_path / "kmer_prgs")
    built = Context(prg, 11, 15, device=-1, from_files=False).export_index()
    loaded = Context(prg, 11, 15, device=-1, from_files=True).export_index()
    for key in built:
        assert np.array_equal(built[key], loaded[key]), key
    # every k-mer node spells a 15-mer whose canonical hash is its index key
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    mask = (1 << 30) - 1
    code = {c: i for i, c in enumerate("ACGT")}
    keys = set(int(x) for x in built["keys"])
    for li, name in enumerate(panel.names):
        text = panel.prgs[li]
        nodes = _gfa_nodes(str(tmp_path / "kmer_prgs" / f"{name}.k15.w11.gfa"))
        assert nodes[0] == [] and nodes[max(nodes)] == []
        for nid, ivs in nodes.items():
            if not ivs:
                continue
            seq = "".join(text[a:b] for a, b in ivs)
            assert len(seq) == 15 and set(seq) <= set("ACGT")
            f = 0
            for ch in seq:
                f = (f << 2) | code[ch]
            rc = seq.encode().translate(comp)[::-1].decode()
            r = 0
            for ch in rc:
                r = (r << 2) | code[ch]
            h = min(int(oracle.lib.orc_hash64(f, mask)), int(oracle.lib.orc_hash64(r, mask)))
            assert h in keys


def test_index_covers_every_minimizer_of_every_haplotype(tmp_path, oracle):
    """Property pinning the PRG sketch to the read sketch: the minimizers of any walk through a PRG are index keys."""
    from drprg_amd import Context, synth
    rng = np.random.default_rng(3)
    for w, k in [(11, 15), (14, 15), (5, 9), (19, 21)]:
        panel = synth.small_panel(seed=w + k, n_loci=3, length=500)
        prg = str(tmp_path / f"p{w}_{k}.prg")
        panel.write(prg)
        idx = Context(prg, w, k, device=-1, from_files=False).export_index()
        keys = set(int(x) for x in idx["keys"])
        for i in range(60):
            hap = synth.sample_haplotype(rng, panel.trees[i % 3])
            h, _, _ = oracle.sketch(hap, w, k)
            assert len(h) > 0 and all(int(x) in keys for x in h)


def test_hot_path_fails_loudly_without_a_device(tmp_path):
    from drprg_amd import Context, DependencyError, synth
    panel = synth.small_panel(seed=1, n_loci=1, length=200)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg)
    ctx = Context(prg, 11, 15, device=-1, from_files=False)
    with pytest.raises(DependencyError) as ei:
        ctx.map_host(np.frombuffer(b"ACGT" * 50, dtype=np.uint8), np.array([0, 200], dtype=np.uint64))
    assert ei.value.code == 19 and "no CPU fallback" in str(ei.value)  # ENODEV
    with pytest.raises(DependencyError):
        ctx.map_fastx(str(tmp_path / "nothing.fq"))


def test_missing_rccl_library_is_enodev_not_a_crash():
    """with no librccl to open, Rccl::get() used to call dlerror() twice (the second call returns NULL once
    the first has cleared the message) and build a std::string from the null pointer -- a crash exactly where the run-time
    binding is supposed to report -ENODEV.  A fresh process (the binding is resolved once per process) with the library name
    pointed at a file that does not exist."""
    code = (
        "import ctypes, sys\n"
        "from drprg_amd._lib import lib\n"
        "ident = (ctypes.c_uint8 * 128)()\n"
        "rc = lib.drprg_hip_comm_unique_id(ident)\n"
        "comm = ctypes.c_void_p()\n"
        "rc2 = lib.drprg_hip_comm_init_rank(ctypes.byref(comm), 1, ident, 0, 0)\n"
        "msg = lib.drprg_hip_last_error(None)\n"
        "print(rc, rc2, (msg or b'').decode())\n")
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, DRPRG_HIP_RCCL_LIB="/nonexistent/librccl-missing.so.1"))
    assert r.returncode == 0, (r.returncode, r.stderr)
    rc, rc2, msg = r.stdout.strip().split(" ", 2)
    assert (int(rc), int(rc2)) == (-19, -19), r.stdout  # -ENODEV from both communicator entries
    assert "librccl-missing.so.1 not found" in msg and "cannot open shared object file" in msg, msg


def test_map_opts_of_another_size_are_refused(tmp_path):
    """drprg_hip_set_opts_sized: a host built against another revision of the header is told so (-EINVAL)."""
    import ctypes as C
    from drprg_amd import Context, synth
    from drprg_amd._lib import MapOpts, lib
    hdr = open(os.path.join(ROOT, "include", "drprg_hip.h")).read()
    assert int(re.search(r"#define DRPRG_HIP_MAP_OPTS_SIZE (\d+)", hdr).group(1)) == C.sizeof(MapOpts)
    panel = synth.small_panel(seed=1, n_loci=1, length=200)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg)
    ctx = Context(prg, 11, 15, device=-1, from_files=False)
    o = MapOpts()
    assert lib.drprg_hip_set_opts_sized(ctx._h, C.byref(o), C.sizeof(o)) == 0
    assert lib.drprg_hip_set_opts_sized(ctx._h, C.byref(o), C.sizeof(o) - 8) == -22
    assert b"mismatch" in lib.drprg_hip_last_error(ctx._h)


def _parse_vcf(path):
    recs = []
    for line in open(path):
        if line.startswith("#"):
            continue
        t = line.rstrip("\n").split("\t")
        fmt = dict(zip(t[8].split(":"), t[9].split(":")))
        recs.append((t, fmt))
    return recs


def test_genotyper_vcf_surface_and_arithmetic(tmp_path, oracle):
    """Host genotyper driven by an oracle-made coverage vector (the GPU is not needed for this stage)."""
    from drprg_amd import Context, synth
    w, k = 11, 15
    panel = synth.small_panel(seed=9)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=20000)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 4000, seed=1)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    out = str(tmp_path / "pandora_genotyped.vcf")
    info = ctx.genotype(genes, out)
    # header: same lines as the reference's raw pandora VCF (date and contigs aside; bcftools' FILTER line aside)
    want = [l for l in open(os.path.join(GOLDEN, "pandora_vcf_surface", "header.vcf")).read().splitlines()
            if not l.startswith(("##fileDate", "##contig", "##FILTER"))]
    got = [l for l in open(out).read().splitlines() if l.startswith("#") and not l.startswith(("##fileDate", "##contig"))]
    assert got == want
    contigs = [l for l in open(out) if l.startswith("##contig")]
    assert len(contigs) == info["loci_present"] == 4
    refs = {n: r for n, r in zip(panel.names, panel.refs)}
    recs = _parse_vcf(out)
    assert len(recs) == info["records"] > 20
    e = info["exp_depth_covg"]
    tot = _present_kmer_totals(ctx, covg)
    zero_thresh = (int(offs[-1]) // 20000) // 10  # (bases mapped / genome size) / 10, as pandora's estimate_parameters
    assert e == int(oracle.lib.orc_exp_depth_covg(tot.ctypes.data, tot.size, zero_thresh))
    seen_classes = set()
    for t, fmt in recs:
        chrom, pos, ref, alts = t[0], int(t[1]), t[3], t[4].split(",")
        assert refs[chrom][pos - 1:pos - 1 + len(ref)] == ref  # REF equals genes.fa at POS (src/consequence.rs:105-113)
        assert alts == sorted(alts) and ref not in alts and "" not in alts and ref != ""
        seen_classes.add(t[7])
        mf = [int(x) for x in fmt["MEAN_FWD_COVG"].split(",")]
        mr = [int(x) for x in fmt["MEAN_REV_COVG"].split(",")]
        gaps = [float(x) for x in fmt["GAPS"].split(",")]
        lik = [float(x) for x in fmt["LIKELIHOOD"].split(",")]
        assert len(mf) == len(alts) + 1
        olik, ogt, oconf = oracle.genotype(mf, mr, gaps, e)
        for a, b in zip(lik, olik):
            assert abs(a - b) <= 1e-5 * max(1, abs(b)) + (e + 1) * 5e-7
        assert int(fmt["GT"]) == ogt
        assert abs(float(fmt["GT_CONF"]) - oconf) <= 2e-5 * max(1, abs(oconf)) + 2 * (e + 1) * 5e-7
    assert any("VC=SNP" in c for c in seen_classes) and any("INDEL" in c for c in seen_classes)
    assert any("NESTED" in c for c in seen_classes) and any("SIMPLE" in c for c in seen_classes)


def _present_kmer_totals(ctx, covg):
    idx = ctx.export_index()
    tot = []
    kb = idx["knode_base"]
    for p in range(ctx.n_prgs):
        c = covg[2 * int(kb[p]):2 * int(kb[p + 1])].reshape(-1, 2).sum(axis=1)
        tot.append(c[1:-1])
    return np.concatenate(tot).astype(np.uint32)


def test_absent_locus_has_no_contig_line(tmp_path, oracle):
    from drprg_amd import Context, synth
    w, k = 11, 15
    panel = synth.small_panel(seed=12)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=20000)
    # reads from locus g2 only
    rng = np.random.default_rng(0)
    hap = np.frombuffer(synth.sample_haplotype(rng, panel.trees[2]).encode(), np.uint8)
    reads = [hap[s:s + 150] for s in rng.integers(0, len(hap) - 150, size=300)]
    offs = np.arange(301, dtype=np.uint64) * np.uint64(150)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(np.concatenate(reads), offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    assert prg_reads[2] > 0 and prg_reads[0] == 0
    ctx.set_coverage(covg, prg_reads, 300 * 150)
    out = str(tmp_path / "o.vcf")
    ctx.genotype(genes, out)
    contigs = [l.strip() for l in open(out) if l.startswith("##contig")]
    assert contigs == ["##contig=<ID=g2>"]  # gene absence is inferred from missing contigs (src/predict.rs:757-765)
    assert {t[0] for t, _ in _parse_vcf(out)} == {"g2"}


def test_cli_index_and_errors(tmp_path):
    from drprg_amd import synth
    panel = synth.small_panel(seed=3, n_loci=2, length=300)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg, str(tmp_path / "genes.fa"))
    r = subprocess.run([PANDORA, "index", "-t", "1", "-w", "11", "-k", "15", prg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(prg + ".k15.w11.idx")
    # missing index files -> non-zero exit, diagnostics on stderr (src/lib.rs:497-506)
    r = subprocess.run([PANDORA, "map", "-w", "10", "-k", "15", "-o", str(tmp_path / "o"), prg, "reads.fq"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "error" in r.stderr
    r = subprocess.run([PANDORA, "index", str(tmp_path / "nope.prg")], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot open" in r.stderr
    r = subprocess.run([PANDORA, "frobnicate"], capture_output=True, text=True)
    assert r.returncode == 2


def test_list_prgs_with_novel_variants_matches_reference_test():
    """/root/reference/src/lib.rs:3009-3050 expects ["gid", "ahpC"] from this file"""
    from drprg_amd import DependencyError, Pandora
    assert Pandora.list_prgs_with_novel_variants(os.path.join(GOLDEN, "denovo_paths_example.txt")) == ["gid", "ahpC"]
    with pytest.raises(DependencyError):
        Pandora.list_prgs_with_novel_variants("/nonexistent/denovo_paths.txt")


def test_synthetic_mtb_like_panel_is_deterministic():
    from drprg_amd import synth
    a, b = synth.mtb_like_panel(), synth.mtb_like_panel()
    assert a.prgs == b.prgs and len(a.prgs) == 18
    assert [len(r) for r in a.refs] == [L for _, L in synth.MTB_LOCI]


@pytest.mark.parametrize("w,k,panel_name", [(11, 15, "mtb"), (14, 15, "small"), (16, 13, "small"), (5, 9, "small"), (1, 15, "small")])
def test_bloom_filters_have_no_false_negatives(tmp_path, w, k, panel_name):
    """The prefiltered kernel may only skip a k-mer its filters reject: every index k-mer (both orientations) must pass
    level 0, levels 1+2 and the second stage of the level-0 form.  Host restatement of the kernels' bit tests
    (PrgIndex::filter_selfcheck), no device needed; the fill bounds keep the filters selective."""
    from drprg_amd import Context, synth
    panel = synth.mtb_like_panel() if panel_name == "mtb" else synth.small_panel(seed=w + k)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg)
    chk = Context(prg, w, k, device=-1, from_files=False).filter_selfcheck()
    assert chk["codes"] > 0
    assert chk["level0_false_negatives"] == 0 and chk["level12_false_negatives"] == 0 and chk["stage2_false_negatives"] == 0
    assert chk["shared_array_false_negatives"] == 0  # level 0 + second-stage bits in one array (second stage inside the streaming kernel); the block filter of its L2 form
    assert 0 < chk["level12_fill_permille"] < 450
    if k == 15:  # level 0 + second stage exist
        assert 0 < chk["level0_fill_permille"] < 300 and 0 < chk["stage2_fill_permille"] < 250
    else:
        assert chk["level0_fill_permille"] == 0 and chk["stage2_fill_permille"] == 0


def test_no_filter_for_wide_kmers(tmp_path):
    from drprg_amd import Context, synth
    panel = synth.small_panel(seed=1)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg)
    assert Context(prg, 19, 21, device=-1, from_files=False).filter_selfcheck()["codes"] == 0


def test_foreign_index_files_are_rebuilt_from_the_prg(tmp_path, capfd):
    """An index directory made by the real pandora has <prg>.kK.wW.idx and kmer_prgs/ in pandora's own format; drprg only
    checks that they exist (/root/reference/src/predict.rs:400-418).  drprg_hip_open then rebuilds the graphs from the
    PRG (same index as a fresh build) unless DRPRG_HIP_STRICT_INDEX is set."""
    from drprg_amd import Context, DependencyError, Pandora, synth
    panel = synth.small_panel(seed=21)
    prg = str(tmp_path / "dr.prg")
    panel.write(prg)
    built = Context(prg, 11, 15, device=-1, from_files=False).export_index()
    # files of the right names, foreign content
    (tmp_path / "kmer_prgs").mkdir()
    (tmp_path / "dr.prg.k15.w11.idx").write_text("1234\n0 1 2 3\n")
    for n in panel.names:
        (tmp_path / "kmer_prgs" / f"{n}.k15.w11.gfa").write_text("H\tVN:Z:1.0\tbn:Z:--linear --singlearr\n")
    loaded = Context(prg, 11, 15, device=-1, from_files=True).export_index()
    assert "rebuilding the k-mer graphs" in capfd.readouterr().err
    for key in built:
        assert np.array_equal(built[key], loaded[key]), key
    os.environ["DRPRG_HIP_STRICT_INDEX"] = "1"
    try:
        with pytest.raises(DependencyError):
            Context(prg, 11, 15, device=-1, from_files=True)
    finally:
        del os.environ["DRPRG_HIP_STRICT_INDEX"]
    # a missing PRG is still an error
    with pytest.raises(DependencyError):
        Context(str(tmp_path / "nope.prg"), 11, 15, device=-1, from_files=True)


# ---- discover (NEXT-2): candidate regions of the called consensus, coverage hand-over to map ---------------------------
def _offpanel_setup(tmp_path, oracle, snp):
    """a 3-locus panel; reads from a sample whose locus g1 carries (snp=True) a substitution that no PRG allele holds, at a
    position well away from any site"""
    from drprg_amd import Context, synth
    w, k = 11, 15
    panel = synth.small_panel(seed=31, n_loci=3, length=900, site_every=60)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    rng = np.random.default_rng(4)
    # the position: middle of the longest site-free stretch of g1's first-allele path
    tree = panel.trees[1]
    pos, cur, best = 0, 0, (0, 0)
    for seg in tree:
        if isinstance(seg, str):
            if len(seg) > best[0]:
                best = (len(seg), cur + len(seg) // 2)
            cur += len(seg)
        else:
            cur += len(seg.alleles[0][0]) if isinstance(seg.alleles[0][0], str) else 0
    pos = best[1]
    assert best[0] > 60
    reads = []
    for locus in range(3):
        hap = synth.sample_haplotype(None, panel.trees[locus], first_allele=True)
        if locus == 1 and snp:
            hap = hap[:pos] + "ACGT".replace(hap[pos], "")[0] + hap[pos + 1:]
        # (the locus sits inside a genome: reads overhang its ends, so coverage does not thin out there)
        h = np.frombuffer((synth.random_seq(rng, 200) + hap + synth.random_seq(rng, 200)).encode(), np.uint8)
        for s in rng.integers(0, len(h) - 150, size=700):
            reads.append(h[s:s + 150])
    offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)
    bases = np.concatenate(reads)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=3000)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    return ctx, genes, pos, panel


def test_discover_locates_an_off_panel_snp_as_a_candidate_region(tmp_path, oracle):
    """/root/reference/src/lib.rs:513-578: the mapping half of discover.  Reads with a SNP the PRG does not hold leave the
    k-mers over it without coverage: exactly one candidate region, on that locus, around that position.  denovo_paths.txt
    still reports 0 loci (no local assembly) in the format list_prgs_with_novel_variants parses (src/lib.rs:648-697)."""
    from drprg_amd import Pandora
    ctx, genes, pos, panel = _offpanel_setup(tmp_path, oracle, snp=True)
    out = tmp_path / "discover"
    out.mkdir()
    regions = ctx.discover(genes, str(out))
    assert [r["locus"] for r in regions] == ["g1"]
    r = regions[0]
    assert r["low_start"] <= pos < r["low_end"] and r["low_end"] - r["low_start"] <= 16
    assert r["start"] == r["low_start"] - 22 and r["end"] == r["low_end"] + 22
    ref = panel.refs[1]
    assert r["seq"] == ref[r["start"]:r["end"]]  # the called consensus here is the reference path
    assert r["max_covg"] <= 3
    assert Pandora.list_prgs_with_novel_variants(str(out / "denovo_paths.txt")) == []
    assert (out / "denovo_sequences.fa").exists()


def test_discover_reports_nothing_for_a_sample_the_panel_explains(tmp_path, oracle):
    ctx, genes, _, _ = _offpanel_setup(tmp_path, oracle, snp=False)
    out = tmp_path / "discover"
    out.mkdir()
    assert ctx.discover(genes, str(out)) == []


def test_low_coverage_interval_rule(tmp_path, oracle):
    """candidate-region arithmetic on a crafted vector: a run of 60 low bases is too long (max 50), runs closer than 22 bases
    merge, uncovered bases never count"""
    from drprg_amd import Context, synth
    w, k = 11, 15
    rng = np.random.default_rng(8)
    seq = synth.random_seq(rng, 1200)
    prg = str(tmp_path / "lin.prg")
    open(prg, "w").write(f">lin\n{seq}\n")
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=1200)
    g = oracle.sketch_prg(seq, w, k, paths=True)
    starts = np.array([p[0][0] for p in g["paths"]])
    n = g["n_nodes"]
    covg = np.zeros(2 * n, np.uint32)
    covg[2::2][:n - 2] = 40  # every k-mer node well covered ...

    def zero(lo, hi):  # ... except those that touch [lo, hi)
        for i, s in enumerate(starts):
            if s < hi and s + k > lo:
                covg[2 * (i + 1)] = 0

    zero(200, 203)
    zero(228, 230)   # within 22 bases of the previous run: merged
    zero(500, 580)   # longer than 50 bases: ignored
    zero(900, 901)
    ctx.set_coverage(covg, np.array([5], np.uint32), 40 * 1200)
    out = tmp_path / "d"
    out.mkdir()
    regions = ctx.discover(None, str(out))
    assert len(regions) == 2
    a, b = regions
    assert a["low_start"] <= 200 and a["low_end"] >= 230 and a["low_end"] - a["low_start"] < 80
    assert b["low_start"] <= 900 < b["low_end"]
    assert all(r["seq"] == seq[r["start"]:r["end"]] for r in regions)


def test_coverage_hand_over_between_discover_and_map(tmp_path, oracle):
    """the vector `discover` saves is taken back only for the same (PRG, reads, parameters) tag"""
    ctx, genes, _, _ = _offpanel_setup(tmp_path, oracle, snp=True)
    cache = str(tmp_path / "cache.bin")
    want = ctx.coverage()
    ctx.save_coverage(cache, "tag-A")
    ctx.set_coverage(np.zeros_like(want[0]), np.zeros_like(want[1]), 0)
    assert not ctx.load_coverage(cache, "tag-B") and ctx.coverage()[0].sum() == 0
    assert not ctx.load_coverage(str(tmp_path / "missing.bin"), "tag-A")
    assert ctx.load_coverage(cache, "tag-A")
    got = ctx.coverage()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    open(cache, "r+b").truncate(100)
    with pytest.raises(Exception):
        ctx.load_coverage(cache, "tag-A")


# ---- discover, second half: novel variants from the reads over the candidate regions, PRG update, re-genotyping ----------
def _denovo_sample(tmp_path, oracle, kind):
    """reads (FASTQ on disk + arrays) of a sample whose locus g1 differs from every PRG path at one place (kind: snp / del / ins),
    well inside a site-free stretch; the locus sits inside flanks so that coverage does not thin out at its ends"""
    from drprg_amd import Context, synth
    w, k = 11, 15
    panel = synth.small_panel(seed=31, n_loci=3, length=900, site_every=60)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    rng = np.random.default_rng(4)
    tree = panel.trees[1]
    cur, best = 0, (0, 0)
    for seg in tree:
        if isinstance(seg, str):
            if len(seg) > best[0]:
                best = (len(seg), cur + len(seg) // 2)
            cur += len(seg)
        else:
            cur += len(seg.alleles[0][0]) if isinstance(seg.alleles[0][0], str) else 0
    pos = best[1]
    ref1 = panel.refs[1]
    if kind == "snp":
        want = (pos, ref1[pos], "ACGT".replace(ref1[pos], "")[0])
        mutated = ref1[:pos] + want[2] + ref1[pos + 1:]
    elif kind == "del":
        want = (pos, ref1[pos:pos + 3], "")
        mutated = ref1[:pos] + ref1[pos + 3:]
    else:
        ins = "ACGTA".replace(ref1[pos], "")[:2] + "T"
        want = (pos, "", ins)
        mutated = ref1[:pos] + ins + ref1[pos:]
    reads = []
    for locus in range(3):
        hap = mutated if locus == 1 else panel.refs[locus]
        h = np.frombuffer((synth.random_seq(rng, 200) + hap + synth.random_seq(rng, 200)).encode(), np.uint8)
        for s in rng.integers(0, len(h) - 150, size=900):
            r = h[s:s + 150]
            reads.append(synth._COMP[r[::-1]] if rng.random() < 0.5 else r)
    offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)
    bases = np.concatenate(reads)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, bases, offs)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=4000)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    return ctx, panel, genes, fq, bases, offs, want


def _same_variant(got, want, ref):
    """(pos, ref, alt) equal up to the left/right shift an indel in a repeat allows"""
    apply = lambda v: ref[:v[0]] + v[2] + ref[v[0] + len(v[1]):]
    return apply(got) == apply(want) and len(got[1]) - len(got[2]) == len(want[1]) - len(want[2])


def _oracle_pile_up(discover_dir, consensus, bases, offs):
    """the oracle's separate statement of the accurate-read pile-up (oracle/oracle_denovo.py) on the candidate regions the product
    wrote and the consensus the test knows the sample to have"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_denovo", os.path.join(ROOT, "oracle", "oracle_denovo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    regions = []
    for line in open(os.path.join(str(discover_dir), "candidate_regions.tsv")):
        if not line.startswith("#"):
            f = line.split("\t")
            regions.append((f[0], int(f[1]), int(f[2])))
    text = bases.tobytes().decode()
    reads = [text[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    return mod.pile_up(consensus, regions, reads)


@pytest.mark.parametrize("kind", ["snp", "del", "ins"])
def test_discover_finds_the_novel_variant_and_the_updated_prg_calls_it(tmp_path, oracle, kind):
    """/root/reference/src/predict.rs:247-302 in one piece: discover -> (update PRG) -> index -> map -> genotype.  The reads carry a
    variant no PRG path holds: discover reports the locus with that variant in denovo_paths.txt (the layout of
    /root/reference/src/lib.rs:3010-3038, parsed by list_prgs_with_novel_variants), the updated PRG holds one more site, and
    mapping the same reads against it calls the new allele (here through the oracle; the GPU e2e test does it on the device)."""
    from drprg_amd import Context, Pandora
    ctx, panel, genes, fq, bases, offs, want = _denovo_sample(tmp_path, oracle, kind)
    out = tmp_path / "discover"
    out.mkdir()
    ctx.set_threads(4)
    variants = ctx.discover_reads(fq, genes, str(out))
    assert len(variants) == 1, variants
    locus, pos1, ref, alt, support, spanning = variants[0]
    assert locus == "g1" and support >= 10 and support >= 0.8 * spanning
    assert _same_variant((pos1 - 1, ref, alt), want, panel.refs[1])
    assert Pandora.list_prgs_with_novel_variants(str(out / "denovo_paths.txt")) == ["g1"]
    assert _oracle_pile_up(out, dict(zip(panel.names, panel.refs)), bases, offs) == [(l, p - 1, r, a, s, n) for l, p, r, a, s, n in variants]
    text = (out / "denovo_paths.txt").read_text()
    assert "\n1 denovo variants for this locus\n" in text and f"\n{pos1}\t{ref}\t{alt}\n" in text
    nodes = re.findall(r"\((\d+) \[(\d+), (\d+)\) ([ACGT]*)\)", text)
    assert len(nodes) >= 3 and all(int(b) - int(a) == len(s) for _, a, b, s in nodes)
    assert "".join(s for *_, s in nodes) == panel.refs[1]  # the called path of a sample that follows the first alleles
    # ---- the PRG with the novel allele as a site ----
    new_prg = str(tmp_path / "updated.dr.prg")
    assert ctx.update_prg(new_prg) == 1
    prgs = [l.rstrip("\n") for l in open(new_prg) if not l.startswith(">")]
    assert prgs[0] == panel.prgs[0] and prgs[2] == panel.prgs[2] and prgs[1] != panel.prgs[1]
    assert len(re.findall(r" (\d+) ", prgs[1])) == len(re.findall(r" (\d+) ", panel.prgs[1])) + 3  # marker, separator, marker
    w, k = 11, 15
    ctx2 = Context(new_prg, w, k, device=-1, from_files=False)
    ctx2.set_opts(illumina=True, genome_size=4000)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx2.set_coverage(covg, prg_reads, int(offs[-1]))
    vcf = str(tmp_path / "updated.vcf")
    ctx2.genotype(genes, vcf)
    called = [(t[0], int(t[1]), t[3], t[4], fmt["GT"]) for t, fmt in _parse_vcf(vcf) if fmt["GT"] not in ("0", ".")]
    assert len(called) == 1 and called[0][0] == "g1"
    _, vpos, vref, valt, gt = called[0]
    assert _same_variant((vpos - 1, vref, valt.split(",")[int(gt) - 1]), want, panel.refs[1])
    # ... and nothing is left to discover on the updated PRG
    out2 = tmp_path / "discover2"
    out2.mkdir()
    assert ctx2.discover_reads(fq, genes, str(out2)) == []


def test_discover_reports_no_novel_variant_for_a_sample_the_panel_explains(tmp_path, oracle):
    ctx, panel, genes, fq, bases, offs, _ = _denovo_sample(tmp_path, oracle, "snp")
    # the same reads against a context whose coverage comes from reference-only reads: no candidate region, nothing assembled
    from drprg_amd import Context, Pandora, synth
    rng = np.random.default_rng(5)
    reads = []
    for locus in range(3):
        h = np.frombuffer((synth.random_seq(rng, 200) + panel.refs[locus] + synth.random_seq(rng, 200)).encode(), np.uint8)
        reads += [h[s:s + 150] for s in rng.integers(0, len(h) - 150, size=700)]
    offs2 = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)
    b2 = np.concatenate(reads)
    fq2 = str(tmp_path / "wt.fq")
    synth.write_fastq(fq2, b2, offs2)
    md, er = map_params(15, True)
    covg, prg_reads, _ = oracle.map_reads(b2, offs2, oracle.build_index(panel.prgs, 11, 15), 11, 15, md, cluster_fraction(er, 15), 10)
    ctx.set_coverage(covg, prg_reads, int(offs2[-1]))
    out = tmp_path / "d"
    out.mkdir()
    assert ctx.discover_reads(fq2, genes, str(out)) == []
    assert Pandora.list_prgs_with_novel_variants(str(out / "denovo_paths.txt")) == []
    assert ctx.update_prg(str(tmp_path / "same.prg")) == 0
    assert [l.rstrip("\n") for l in open(tmp_path / "same.prg") if not l.startswith(">")] == panel.prgs


@pytest.mark.parametrize("kind", ["snp", "del", "ins"])
def test_discover_with_noisy_long_reads(tmp_path, oracle, kind):
    """the same off-panel variant under Nanopore-like reads (5 % errors, no -I): the reads' strings between the anchors are
    aligned to the consensus and counted column by column; exactly the planted variant comes out, and no read error does"""
    from types import SimpleNamespace
    from drprg_amd import Context, synth
    ctx_i, panel, genes, _, _, _, want = _denovo_sample(tmp_path, oracle, kind)
    ref1 = panel.refs[1]
    mutated = ref1[:want[0]] + want[2] + ref1[want[0] + len(want[1]):]
    rng = np.random.default_rng(8)
    parts = [synth.random_seq(rng, 300)]
    for locus in range(3):
        parts += [mutated if locus == 1 else panel.refs[locus], synth.random_seq(rng, 300)]
    genome = np.frombuffer("".join(parts).encode(), np.uint8)
    g = SimpleNamespace(haps=[genome], lens=np.array([genome.size], dtype=np.int64))
    bases, offs = synth.sample_long_reads(g, 45 * genome.size // 1500, seed=12, mean_len=1500, min_len=400, max_len=genome.size - 1)
    fq = str(tmp_path / "long.fq")
    synth.write_fastq(fq, bases, offs)
    w, k = 11, 15
    ctx = Context(str(tmp_path / "dr.prg"), w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=False, genome_size=int(genome.size))
    md, er = map_params(k, False)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    out = tmp_path / "discover_long"
    out.mkdir()
    ctx.set_threads(4)
    variants = ctx.discover_reads(fq, genes, str(out))
    assert len(variants) == 1, variants
    locus, pos1, ref, alt, support, spanning = variants[0]
    assert locus == "g1" and support >= 4 and spanning >= support
    assert _same_variant((pos1 - 1, ref, alt), want, ref1)
    # the oracle's own statement of the column vote (oracle/oracle_denovo.py column_vote: whole reads, str.find, its own alignment):
    # the same variant, support and spanning count
    from util import oracle_denovo
    assert oracle_denovo(out, dict(zip(panel.names, panel.refs)), bases, offs, noisy=True) == [(l, p - 1, r, a, s, n) for l, p, r, a, s, n in variants]


def test_discover_lists_several_loci_several_variants_and_a_variant_inside_a_nested_allele(tmp_path, oracle):
    """NEXT-2 at drop-in quality: a denovo_paths.txt with three loci -- two novel SNPs in one locus, one in another, one inside the
    long allele of a NESTED site of a third -- read back by the line-for-line port of the reference's parser
    (/root/reference/src/lib.rs:648-697: the `<N> loci with denovo variants` line, a locus name before every `<n> nodes` line).
    The PRG update places all four: the one in the nested allele becomes a site inside that allele (never somewhere else), a
    fifth variant that overlaps an existing site is reported and skipped; mapping against the updated PRG calls every placed allele
    at its own position."""
    from drprg_amd import Context, Pandora, synth
    from drprg_amd.synth import Site
    rng = np.random.default_rng(77)
    rs = lambda n: synth.random_seq(rng, n)
    w, k = 11, 15
    # g0: two site-free stretches around one SNP site; g1: a nested site whose second allele is long; g2: plain
    t0 = [rs(260), Site([["A"], ["C"]]), rs(260)]
    inner = Site([["G"], ["T"]])
    long_allele = [rs(14), inner, rs(40)]
    t1 = [rs(250), Site([["C"], long_allele]), rs(250)]
    t2 = [rs(240), Site([["T"], ["G"]]), rs(240), Site([["AC"], ["A"]]), rs(60)]
    panel = synth.Panel(["g0", "g1", "g2"], [t0, t1, t2])
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)

    def flip(s, i):
        return s[:i] + "ACGT".replace(s[i], "")[0] + s[i + 1:]

    # the sample: g0 follows the first alleles with novel SNPs at 100 and 400; g1 takes the long allele (inner G) with a novel SNP in
    # the 40-base stretch behind the inner site; g2 has a novel SNP at 120 and one that overlaps its first site (position 240)
    h0 = flip(flip(panel.refs[0], 100), 400)
    g1_hap = t1[0] + long_allele[0] + "G" + long_allele[2] + t1[2]
    p1 = len(t1[0]) + len(long_allele[0]) + 1 + 20
    h1 = flip(g1_hap, p1)
    h2 = flip(panel.refs[2], 120)
    h2 = h2[:239] + "ACGT".replace(h2[239], "")[1] + "ACGT".replace(h2[240], "").replace("G", "")[0] + h2[241:]  # 239-240: across the T/G site
    reads = []
    for hap in (h0, h1, h2):
        h = np.frombuffer((rs(200) + hap + rs(200)).encode(), np.uint8)
        for s in rng.integers(0, len(h) - 150, size=1100):
            r = h[s:s + 150]
            reads.append(synth._COMP[r[::-1]] if rng.random() < 0.5 else r)
    offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)
    bases = np.concatenate(reads)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, bases, offs)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=3000)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    out = tmp_path / "discover"
    out.mkdir()
    ctx.set_threads(4)
    variants = ctx.discover_reads(fq, genes, str(out))
    by_locus = {}
    for locus, pos1, ref, alt, support, spanning in variants:
        by_locus.setdefault(locus, []).append((pos1, ref, alt))
    assert sorted(by_locus) == ["g0", "g1", "g2"], variants
    assert [p for p, _, _ in by_locus["g0"]] == [101, 401] and all(len(r) == len(a) == 1 for _, r, a in by_locus["g0"])
    assert by_locus["g1"] == [(p1 + 1, g1_hap[p1], h1[p1])]
    assert (121, panel.refs[2][120], h2[120]) in by_locus["g2"]
    # the oracle's own pile-up over the same regions agrees variant for variant, support for support
    want = _oracle_pile_up(out, {"g0": panel.refs[0], "g1": g1_hap, "g2": panel.refs[2]}, bases, offs)
    assert want == [(l, p - 1, r, a, s, n) for l, p, r, a, s, n in variants]
    # ---- the file the reference parses ----
    path = str(out / "denovo_paths.txt")
    assert Pandora.list_prgs_with_novel_variants(path) == ["g0", "g1", "g2"]
    text = open(path).read()
    assert text.startswith("1 samples\nSample sample\n3 loci with denovo variants\n")
    blocks = re.split(r"\n(?=g[012]\n\d+ nodes\n)", text)
    assert len(blocks) == 4
    for name, blk in zip(("g0", "g1", "g2"), blocks[1:]):
        lines = blk.split("\n")
        assert lines[0] == name
        n_nodes = int(lines[1].split()[0])
        nodes = [re.fullmatch(r"\((\d+) \[(\d+), (\d+)\) ([ACGT]*)\)", l) for l in lines[2:2 + n_nodes]]
        assert all(nodes) and all(int(m.group(3)) - int(m.group(2)) == len(m.group(4)) for m in nodes)
        # the node intervals are offsets into the PRG string (markers and their spaces included)
        pstr = panel.prgs[("g0", "g1", "g2").index(name)]
        assert all(pstr[int(m.group(2)):int(m.group(3))] == m.group(4) for m in nodes)
        nv = int(lines[2 + n_nodes].split()[0])
        assert lines[2 + n_nodes] == f"{nv} denovo variants for this locus" and nv == len(by_locus[name])
        got = [tuple(l.split("\t")) for l in lines[3 + n_nodes:3 + n_nodes + nv]]
        assert got == [(str(p), r, a) for p, r, a in by_locus[name]]
    g1_nodes = "".join(m.group(4) for m in [re.fullmatch(r"\((\d+) \[(\d+), (\d+)\) ([ACGT]*)\)", l) for l in blocks[2].split("\n")[2:]] if m)
    assert g1_nodes == g1_hap  # the called path runs through the long (nested) allele
    # ---- the update: all placed -- the one across the existing site takes that site along as the first allele of a new site (round 4;
    # rounds 2-3 reported it and left it out) ----
    new_prg = str(tmp_path / "updated.dr.prg")
    n_sites_before = [len(re.findall(r" \d+ ", p)) for p in panel.prgs]
    applied = ctx.update_prg(new_prg)
    placed = sum(len(v) for v in by_locus.values())
    assert applied == placed and applied >= 5
    prgs = [l.rstrip("\n") for l in open(new_prg) if not l.startswith(">")]
    # every updated PRG parses the way pandora parses (markers numbered in the order its parser meets the sites), spells everything the
    # old one spelled, and spells the sample's haplotype
    from util import prg_language
    for old, new, hap in zip(panel.prgs, prgs, (h0, h1, h2)):
        lang = prg_language(new)
        assert prg_language(old) <= lang and hap in lang and hap not in prg_language(old)
    assert re.search(r" 7 [ACGT]* ?9 T 10 G 9  8 [ACGT]+ 7 ", prgs[2]), prgs[2][200:320]  # the T/G site inside the new site's first allele
    # g1: the new site sits inside the long allele, i.e. between the inner site's close and the outer site's close
    m = re.search(r" 7 (?P<tail>[ACGT]+ \d+ [ACGT] \d+ [ACGT] \d+ [ACGT]+) 5 ", prgs[1])
    assert m, prgs[1][240:420]
    assert len(re.findall(r" \d+ ", prgs[1])) == n_sites_before[1] + 3
    # ---- every placed allele is called at its position on the updated PRG ----
    ctx2 = Context(new_prg, w, k, device=-1, from_files=False)
    ctx2.set_opts(illumina=True, genome_size=3000)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx2.set_coverage(covg, prg_reads, int(offs[-1]))
    vcf = str(tmp_path / "updated.vcf")
    ctx2.genotype(genes, vcf)
    alt_calls = {(t[0], int(t[1])) for t, fmt in _parse_vcf(vcf) if fmt["GT"] not in ("0", ".")}
    assert ("g0", 101) in alt_calls and ("g0", 401) in alt_calls and ("g2", 121) in alt_calls
    assert any(c == "g1" for c, _ in alt_calls)
    assert any(c == "g2" and 238 <= p <= 242 for c, p in alt_calls), alt_calls  # the allele across the old site


def test_named_index_with_foreign_files_is_an_error_unless_rebuild_is_asked_for(tmp_path):
    """NEXT-3: `-x mtb` resolves a downloaded index (~/.drprg/mtb/mtb-<version>, /root/reference/src/cli.rs:21-78) whose
    dr.prg.k15.w11.idx / kmer_prgs are the real pandora's files.  If they do not parse in the layout this build reads, a silent
    rebuild from dr.prg would hide that: `drprg predict` stops there; --rebuild-index (or DRPRG_HIP_REBUILD_INDEX=1) asks for the
    rebuild; an index given as a path keeps the rebuild with a warning."""
    from drprg_amd import synth
    exe = os.path.join(ROOT, "drprg_amd", "bin", "drprg")
    home = tmp_path / "home"
    idx = home / ".drprg" / "mtb" / "mtb-20230308"
    (idx / "kmer_prgs").mkdir(parents=True)
    (idx / "msas").mkdir()
    panel = synth.small_panel(seed=3)
    panel.write(str(idx / "dr.prg"), str(idx / "genes.fa"))
    (idx / "dr.prg.k15.w11.idx").write_text("1234\n0 1 2 3\n")  # files of the right names, foreign content
    for n in panel.names:
        (idx / "kmer_prgs" / f"{n}.k15.w11.gfa").write_text("H\tVN:Z:1.0\tbn:Z:--linear --singlearr\n")
    (idx / ".config.toml").write_text('min_match_len = 5\nmax_nesting = 5\nk = 15\nw = 11\npadding = 100\nversion = "20230308"\n')
    for f in ("panel.bcf", "panel.bcf.csi"):
        (idx / f).write_bytes(open(os.path.join(GOLDEN, "downstream", f), "rb").read())
    reads = tmp_path / "r.fq"
    reads.write_text("@r\nACGT\n+\nIIII\n")
    env = dict(os.environ, HOME=str(home))
    env.pop("DRPRG_HIP_REBUILD_INDEX", None)
    env.pop("DRPRG_HIP_STRICT_INDEX", None)
    r = subprocess.run([exe, "predict", "-x", "mtb", "-i", str(reads), "-o", str(tmp_path / "o1")], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "cannot open the index" in r.stderr and "--rebuild-index" in r.stderr and "rebuilding the" not in r.stderr, r.stderr
    assert ".idx" in r.stderr or "kmer_prgs" in r.stderr
    r = subprocess.run([exe, "predict", "-x", "mtb@20230308", "-i", str(reads), "-o", str(tmp_path / "o2"), "--rebuild-index"],
                       capture_output=True, text=True, env=env)
    assert "rebuilding the k-mer graphs" in r.stderr, r.stderr  # (then it needs a GPU: -ENODEV on this host)
    r = subprocess.run([exe, "predict", "-x", str(idx), "-i", str(reads), "-o", str(tmp_path / "o3")], capture_output=True, text=True, env=env)
    assert "rebuilding the k-mer graphs" in r.stderr, r.stderr


def test_package_workload_data_equals_the_golden_fixtures():
    """drprg_amd/data/mtb_8d holds the two data files the SURVEY-8d workload is built from (so that bench.py does not depend on
    tests/): the same bytes as the fixtures tools/make_golden.py copied from /root/reference/tests/cases/predict"""
    import filecmp
    from drprg_amd import synth
    for f in ("genes.fa", "panel.bcf"):
        assert filecmp.cmp(os.path.join(synth.MTB_8D_DIR, f), os.path.join(GOLDEN, "downstream", f), shallow=False), f
    assert len(synth.mtb_8d_panel().names) == 18
