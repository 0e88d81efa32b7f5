"""Reads that stay in HBM (include/drprg_hip.h: drprg_hip_keep_reads / drprg_hip_map_resident): the second walk over the reads that
`pandora discover` does, and the `pandora map` drprg runs on the updated PRG (/root/reference/src/predict.rs:248-256, :296-302),
without a second pass over the file.  The bar: the same bytes out as when the file is read again."""
import os
import subprocess

import numpy as np
import pytest

from util import ROOT

pytestmark = pytest.mark.gpu
W, K = 11, 15


def _sample(tmp_path, n_background=0, kind="snp", noisy=False, odd_bases=False, seed=4):
    """three loci with sites, locus g1 of the sample carrying one off-panel change; optionally reads of an unrelated background
    genome in between (they hold no anchor), lower-case stretches and N in some reads"""
    from drprg_amd import synth
    panel = synth.small_panel(seed=31, n_loci=3, length=900, site_every=60)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    rng = np.random.default_rng(seed)
    # the middle of the longest site-free stretch of g1
    cur, best = 0, (0, 0)
    for seg in panel.trees[1]:
        if isinstance(seg, str):
            if len(seg) > best[0]:
                best = (len(seg), cur + len(seg) // 2)
            cur += len(seg)
        else:
            cur += len(seg.alleles[0][0]) if isinstance(seg.alleles[0][0], str) else 0
    pos, ref1 = best[1], panel.refs[1]
    if kind == "snp":
        mutated = ref1[:pos] + "ACGT".replace(ref1[pos], "")[0] + ref1[pos + 1:]
    elif kind == "del":
        mutated = ref1[:pos] + ref1[pos + 3:]
    else:
        mutated = ref1[:pos] + "ACGTA".replace(ref1[pos], "")[:2] + "T" + ref1[pos:]
    reads = []
    if noisy:
        from types import SimpleNamespace
        for locus in range(3):
            hap = mutated if locus == 1 else panel.refs[locus]
            h = np.frombuffer((synth.random_seq(rng, 400) + hap + synth.random_seq(rng, 400)).encode(), np.uint8)
            b, o = synth.sample_long_reads(SimpleNamespace(haps=[h], lens=np.array([h.size], dtype=np.int64)), 60, seed=seed + locus, mean_len=900, min_len=400,
                                           max_len=h.size - 1)
            reads += [b[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]
    else:
        for locus in range(3):
            hap = mutated if locus == 1 else panel.refs[locus]
            h = np.frombuffer((synth.random_seq(rng, 200) + hap + synth.random_seq(rng, 200)).encode(), np.uint8)
            for s in rng.integers(0, len(h) - 150, size=900):
                r = h[s:s + 150]
                reads.append(synth._COMP[r[::-1]] if rng.random() < 0.5 else r)
    if n_background:
        g = np.frombuffer(synth.random_seq(rng, 200000).encode(), np.uint8)
        starts = rng.integers(0, g.size - 150, size=n_background)
        reads += list(g[starts[:, None] + np.arange(150)])
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    if odd_bases:
        for i in range(0, len(reads), 7):
            r = reads[i].copy()
            if i % 14 == 0:
                r[int(rng.integers(0, r.size))] = ord("N")
            else:
                a = int(rng.integers(0, r.size - 20))
                r[a:a + 20] |= 0x20  # lower case
            reads[i] = r
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([r.size for r in reads])
    bases = np.concatenate(reads)
    fq = str(tmp_path / "reads.fq")
    if all(r.size == 150 for r in reads):
        synth.write_fastq_fixed(fq, bases, 150)
    else:
        synth.write_fastq(fq, bases, offs)
    return panel, prg, genes, fq


def _fastq_arrays(fq):
    seqs = [line.strip() for i, line in enumerate(open(fq)) if i % 4 == 1]
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in seqs])
    return np.frombuffer("".join(seqs).encode(), np.uint8), offs


def _files(d):
    return {f: open(os.path.join(str(d), f), "rb").read() for f in ("candidate_regions.tsv", "denovo_variants.tsv", "denovo_paths.txt", "denovo_sequences.fa")}


def _discover(prg, genes, fq, out, keep, illumina=True, threads=4, packed=False):
    from drprg_amd import Context
    out.mkdir()
    ctx = Context(prg, W, K, device=0, from_files=False)
    ctx.set_opts(illumina=illumina, genome_size=4000)
    ctx.set_threads(threads)
    ctx.set_input_format(packed)
    if keep:
        ctx.keep_reads(keep)
    ctx.map_fastx(fq)
    variants = ctx.discover_reads(fq, genes, str(out))
    return ctx, variants


@pytest.mark.parametrize("kind,noisy,odd", [("snp", False, False), ("del", False, True), ("ins", False, False), ("snp", True, False), ("del", True, False)])
def test_discover_from_resident_reads_writes_what_discover_from_the_file_writes(tmp_path, kind, noisy, odd):
    panel, prg, genes, fq = _sample(tmp_path, kind=kind, noisy=noisy, odd_bases=odd)
    a, va = _discover(prg, genes, fq, tmp_path / "file", 0, illumina=not noisy)
    b, vb = _discover(prg, genes, fq, tmp_path / "hbm", 1 << 30, illumina=not noisy)
    assert not a.resident_info()["last_discover_from_hbm"] and not a.resident_info()["complete"]
    info = b.resident_info()
    assert info["complete"] and info["last_discover_from_hbm"] and info["blocks"] >= 1 and info["bytes"] > 0
    assert len(va) == 1 and va == vb
    assert _files(tmp_path / "file") == _files(tmp_path / "hbm")
    assert np.array_equal(a.coverage()[0], b.coverage()[0]) and a.counters() == b.counters()
    # ... and what came out of the anchor-scan path is what the ORACLE's separate statement of the pile-up finds in the whole read set
    # (oracle/oracle_denovo.py: pile_up for accurate reads, column_vote for noisy ones -- whole reads and str.find, no k-mer tables, no
    # device): the same variant with the same support and spanning counts (f-2's GPU tests compared the HIP path with itself)
    from util import oracle_denovo
    bases, offs = _fastq_arrays(fq)
    want = oracle_denovo(tmp_path / "hbm", dict(zip(panel.names, panel.refs)), bases, offs, noisy=noisy)
    assert want == [(l, p - 1, r, alt, s, n) for l, p, r, alt, s, n in vb]
    # ... and with the reads kept in HBM in the 2-bit packed form (a quarter of the device memory): the same files
    c, vc = _discover(prg, genes, fq, tmp_path / "hbm_packed", 1 << 30, illumina=not noisy, packed=True)
    assert c.resident_info()["last_discover_from_hbm"] and c.resident_info()["bytes"] < info["bytes"]
    assert vc == vb and np.array_equal(c.coverage()[0], b.coverage()[0])
    fa, fc = _files(tmp_path / "hbm"), _files(tmp_path / "hbm_packed")
    if odd:  # (reads with lower-case stretches come back upper-cased from a packed block: the sequences file may differ in case only)
        fa = {k: v.upper() for k, v in fa.items()}
        fc = {k: v.upper() for k, v in fc.items()}
    assert fa == fc


def test_local_assembly_from_resident_reads(tmp_path, oracle):
    """round 4: the local (de Bruijn) assembly piles up every read that shares a k-mer with a region's slice of the consensus; with the reads
    in HBM the device selects them (all the slice's k-mers go to anchor_scan_kernel, not two anchors per region).  100-base reads and a
    24-base insertion: no read holds both anchors, the pile-up finds nothing, the assembly does -- the same from the file, from HBM, from
    packed blocks in HBM, and by the oracle's own statement of the assembly (oracle/oracle_denovo.py local_assembly)"""
    from drprg_amd import Context
    from test_local_assembly import _sample as assembly_sample
    from util import oracle_local_assembly
    host, panel, genes, fq, bases, offs, hs, pos = assembly_sample(tmp_path, oracle, lambda ref, pos: [(1.0, ref[:pos] + "TCGGATACCGTTAGCAATGCCTAG" + ref[pos + 2:])],
                                                                 read_len=100, n_reads=1400)
    prg = str(tmp_path / "dr.prg")
    a, va = _discover(prg, genes, fq, tmp_path / "file", 0)
    b, vb = _discover(prg, genes, fq, tmp_path / "hbm", 1 << 30)
    c, vc = _discover(prg, genes, fq, tmp_path / "hbm_packed", 1 << 30, packed=True)
    assert b.resident_info()["last_discover_from_hbm"] and c.resident_info()["last_discover_from_hbm"] and not a.resident_info()["last_discover_from_hbm"]
    assert len(va) == 1 and va == vb == vc and va[0][0] == "g1"
    mutated = panel.refs[1][:va[0][1] - 1] + va[0][3] + panel.refs[1][va[0][1] - 1 + len(va[0][2]):]
    assert mutated == hs[0][1]
    assert oracle_local_assembly(tmp_path / "hbm", dict(zip(panel.names, panel.refs)), bases, offs) == [(l, p - 1, r, alt, s, n) for l, p, r, alt, s, n in vb]
    assert _files(tmp_path / "file") == _files(tmp_path / "hbm")


def test_many_blocks_and_reads_without_anchors(tmp_path):
    """300 k reads, most of them from an unrelated genome, eight parser threads -> several blocks in HBM; anchors that run over the
    end of a read into the next read of the block select a read too many at worst"""
    panel, prg, genes, fq = _sample(tmp_path, n_background=300000, odd_bases=True)
    a, va = _discover(prg, genes, fq, tmp_path / "file", 0, threads=8)
    b, vb = _discover(prg, genes, fq, tmp_path / "hbm", 1 << 30, threads=8)
    assert b.resident_info()["last_discover_from_hbm"] and b.resident_info()["blocks"] >= 2
    assert len(va) == 1 and va == vb and _files(tmp_path / "file") == _files(tmp_path / "hbm")
    from util import oracle_denovo
    bases, offs = _fastq_arrays(fq)
    assert oracle_denovo(tmp_path / "hbm", dict(zip(panel.names, panel.refs)), bases, offs) == [(l, p - 1, r, alt, s, n) for l, p, r, alt, s, n in vb]


def test_mapping_the_resident_reads_against_the_updated_prg(tmp_path):
    from drprg_amd import Context
    panel, prg, genes, fq = _sample(tmp_path, n_background=50000)
    b, vb = _discover(prg, genes, fq, tmp_path / "hbm", 1 << 30, threads=8)
    new_prg = str(tmp_path / "updated.dr.prg")
    assert b.update_prg(new_prg) == 1
    got = Context(new_prg, W, K, device=0, from_files=False)
    got.set_opts(illumina=True, genome_size=4000)
    got.map_resident(b)
    want = Context(new_prg, W, K, device=0, from_files=False)
    want.set_opts(illumina=True, genome_size=4000)
    want.set_threads(3)
    want.map_fastx(fq)
    assert np.array_equal(got.coverage()[0], want.coverage()[0]) and np.array_equal(got.coverage()[1], want.coverage()[1])
    cg, cw = got.counters(), want.counters()
    assert cg == cw and cg["reads"] == 50000 + 2700
    got.genotype(genes, str(tmp_path / "got.vcf"))
    want.genotype(genes, str(tmp_path / "want.vcf"))
    assert open(tmp_path / "got.vcf").read() == open(tmp_path / "want.vcf").read()
    # the source context is unchanged by the call: a second target gets the same
    again = Context(new_prg, W, K, device=0, from_files=False)
    again.set_opts(illumina=True, genome_size=4000)
    again.map_resident(b)
    assert np.array_equal(again.coverage()[0], want.coverage()[0])


def test_the_limit_and_everything_that_makes_the_resident_reads_stand_for_something_else(tmp_path):
    from drprg_amd import Context
    from drprg_amd.pandora import DependencyError
    panel, prg, genes, fq = _sample(tmp_path)
    ref_ctx, want = _discover(prg, genes, fq, tmp_path / "file", 0)
    # a limit the sample does not fit in: nothing is kept, discover reads the file, the same comes out
    small, got = _discover(prg, genes, fq, tmp_path / "small", 1000)
    info = small.resident_info()
    assert not info["complete"] and info["bytes"] == 0 and not info["last_discover_from_hbm"]
    assert got == want and _files(tmp_path / "small") == _files(tmp_path / "file")
    other = Context(prg, W, K, device=0, from_files=False)
    other.set_opts(illumina=True, genome_size=4000)
    with pytest.raises(DependencyError) as e:
        other.map_resident(small)
    assert e.value.code == 61
    with pytest.raises(DependencyError):
        other.map_resident(other)
    # two files in one context: the reads in HBM are not "this file"
    two = Context(prg, W, K, device=0, from_files=False)
    two.set_opts(illumina=True, genome_size=4000)
    two.keep_reads(1 << 30)
    two.map_fastx(fq)
    fq2 = str(tmp_path / "again.fq")
    os.link(fq, fq2)
    two.map_fastx(fq2)
    (tmp_path / "two").mkdir()
    two.discover_reads(fq, genes, str(tmp_path / "two"))
    assert two.resident_info()["complete"] and not two.resident_info()["last_discover_from_hbm"]
    # ... but they are still everything the context mapped: another index can map them
    other.map_resident(two)
    assert other.counters()["reads"] == 2 * 2700
    # reset drops them
    two.reset()
    assert two.resident_info()["bytes"] == 0
    two.map_fastx(fq)
    (tmp_path / "three").mkdir()
    assert two.discover_reads(fq, genes, str(tmp_path / "three")) == want and two.resident_info()["last_discover_from_hbm"]
    # a batch mapped from the caller's own buffers is not among the kept ones
    bases = np.frombuffer(b"ACGT" * 50, np.uint8)
    two.map_host(bases, np.array([0, 200], dtype=np.uint64))
    assert not two.resident_info()["complete"]


def test_keep_reads_switched_on_after_reads_were_mapped(tmp_path):
    """keep_reads on a context that has already mapped reads -- the earlier reads are not in HBM, so the context does
    not hold "every read mapped since the last reset" and map_resident must refuse (-ENODATA) instead of mapping a subset."""
    from drprg_amd import Context
    from drprg_amd.pandora import DependencyError
    panel, prg, genes, fq = _sample(tmp_path)
    late = Context(prg, W, K, device=0, from_files=False)
    late.set_opts(illumina=True, genome_size=4000)
    late.map_fastx(fq)          # mapped, not kept
    late.keep_reads(1 << 30)
    late.map_fastx(fq)          # mapped and kept
    info = late.resident_info()
    assert not info["complete"]
    other = Context(prg, W, K, device=0, from_files=False)
    other.set_opts(illumina=True, genome_size=4000)
    with pytest.raises(DependencyError) as e:
        other.map_resident(late)
    assert e.value.code == 61 and other.counters()["reads"] == 0
    # after a reset the same context keeps everything it maps
    late.reset()
    late.map_fastx(fq)
    assert late.resident_info()["complete"]
    other.map_resident(late)
    assert other.counters()["reads"] == 2700


def test_pandora_discover_takes_its_second_pass_from_hbm(tmp_path):
    """the `pandora discover` executable drprg spawns: same files with DRPRG_HIP_KEEP_READS_GB=0 (the file is read twice)"""
    panel, prg, genes, fq = _sample(tmp_path, n_background=20000)
    exe = os.path.join(ROOT, "drprg_amd", "bin", "pandora")
    r = subprocess.run([exe, "index", "-t", "4", "-w", str(W), "-k", str(K), prg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    q = tmp_path / "query.tsv"
    q.write_text(f"s\t{fq}\n")
    outs = {}
    for name, gb in (("hbm", None), ("file", "0")):
        env = dict(os.environ)
        if gb is not None:
            env["DRPRG_HIP_KEEP_READS_GB"] = gb
        r = subprocess.run([exe, "discover", "-t", "4", "-w", str(W), "-k", str(K), "-I", "-g", "4000", "-o", str(tmp_path / name), prg, str(q)],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert ("resident in device memory" in r.stdout) == (gb is None), r.stdout
        outs[name] = _files(tmp_path / name)
    assert outs["hbm"] == outs["file"] and b"1 denovo variants" in outs["hbm"]["denovo_paths.txt"]


def test_resident_reads_of_a_context_over_several_devices(tmp_path):
    """drprg_hip_open_multi (here: device 0 listed twice, two Mapper objects): every device keeps the blocks it mapped, discover asks
    all of them, and the context of the updated PRG maps each device's blocks on that device"""
    from drprg_amd import Context
    panel, prg, genes, fq = _sample(tmp_path, n_background=120000)
    one, want = _discover(prg, genes, fq, tmp_path / "one", 0, threads=8)
    two = Context(prg, W, K, from_files=False, devices=[0, 0])
    two.set_opts(illumina=True, genome_size=4000)
    two.set_threads(8)
    two.keep_reads(1 << 30)
    two.map_fastx(fq)
    (tmp_path / "two").mkdir()
    got = two.discover_reads(fq, genes, str(tmp_path / "two"))
    info = two.resident_info()
    assert info["complete"] and info["last_discover_from_hbm"] and info["blocks"] >= 2
    assert got == want and _files(tmp_path / "two") == _files(tmp_path / "one")
    assert np.array_equal(two.coverage()[0], one.coverage()[0])
    new_prg = str(tmp_path / "updated.dr.prg")
    assert two.update_prg(new_prg) == 1
    nxt = Context(new_prg, W, K, from_files=False, devices=[0, 0])
    nxt.set_opts(illumina=True, genome_size=4000)
    nxt.map_resident(two)
    ref = Context(new_prg, W, K, device=0, from_files=False)
    ref.set_opts(illumina=True, genome_size=4000)
    ref.set_threads(4)
    ref.map_fastx(fq)
    assert np.array_equal(nxt.coverage()[0], ref.coverage()[0]) and np.array_equal(nxt.coverage()[1], ref.coverage()[1])
    assert nxt.counters()["reads"] == ref.counters()["reads"] == 120000 + 2700
    # a context over one device cannot take the reads of one over two
    single = Context(new_prg, W, K, device=0, from_files=False)
    with pytest.raises(Exception):
        single.map_resident(two)
