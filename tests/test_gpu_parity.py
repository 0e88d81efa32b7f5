"""HIP path vs CPU oracle, bit-exact, through the C ABI (pytest -m gpu).

Both forms of the sketch kernel are checked: kernel=1 is the direct sketch (every k-mer hashed), kernel=2 the
persistent Bloom-prefiltered form (k <= 15, w <= 16); they must give the identical coverage vector."""
import numpy as np
import pytest

from util import cluster_fraction, map_params

pytestmark = pytest.mark.gpu

import os

# DRPRG_FT_DEBUG=8 sends every read through the generic pipeline (a second way to run this whole file): the counts of
# reads the per-read kernel left over mean nothing then
FORCED_GENERIC = bool(int(os.environ.get("DRPRG_FT_DEBUG", "0") or 0) & 8)


def _experimental():
    from drprg_amd import _lib
    return bool(_lib.lib.drprg_hip_experimental())


EXPERIMENTAL = _experimental()  # the library holds the opt-in kernel forms (make EXPERIMENTAL=1, DRPRG_HIP_LIB)


def _ctx(tmp_path, panel, w, k, illumina, genome_size=20000, kernel=0, min_cluster_size=10):
    from drprg_amd import Context
    prg = str(tmp_path / "dr.prg")
    panel.write(prg, str(tmp_path / "genes.fa"))
    ctx = Context(prg, w, k, device=0, from_files=False, threads=8)
    ctx.set_opts(illumina=illumina, genome_size=genome_size, kernel=kernel, min_cluster_size=min_cluster_size)
    ctx.prg_strings = panel.prgs  # (what the oracle builds its own index from)
    return ctx


_ORACLE_INDEX = {}


def _oracle_index(oracle, prg_strings, w, k):
    """the oracle's own index of these PRG strings (oracle/oracle_index.c): nothing but the reads is shared with the product"""
    key = (hash(tuple(prg_strings)), w, k)
    if key not in _ORACLE_INDEX:
        if len(_ORACLE_INDEX) > 4:
            _ORACLE_INDEX.clear()
        _ORACLE_INDEX[key] = oracle.build_index(prg_strings, w, k)
    return _ORACLE_INDEX[key]


def _oracle_map(oracle, idx, bases, offsets, w, k, illumina, min_cluster_size=10, threads=1):
    """oracle.map_reads, optionally over disjoint read ranges on several threads (the C call releases the GIL; integer
    coverage sums commute)"""
    md, er = map_params(k, illumina)
    frac = cluster_fraction(er, k)
    n = len(offsets) - 1
    if threads <= 1 or n < 4 * threads:
        return oracle.map_reads(bases, offsets, idx, w, k, md, frac, min_cluster_size)
    from concurrent.futures import ThreadPoolExecutor
    offsets = np.ascontiguousarray(offsets, np.uint64)
    cuts = [n * i // threads for i in range(threads + 1)]

    def part(i):
        lo, hi = cuts[i], cuts[i + 1]
        return oracle.map_reads(bases[int(offsets[lo]):int(offsets[hi])], offsets[lo:hi + 1] - offsets[lo], idx, w, k, md, frac,
                                min_cluster_size)

    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(part, range(threads)))
    cov = np.sum(np.stack([p[0] for p in parts]).astype(np.uint64), axis=0).astype(np.uint32)
    prg = np.sum(np.stack([p[1] for p in parts]).astype(np.uint64), axis=0).astype(np.uint32)
    cnt = {key: sum(p[2][key] for p in parts) for key in parts[0][2]}
    return cov, prg, cnt


def _compare(ctx, oracle, bases, offsets, w, k, illumina, kernel, min_cluster_size=10, threads=1):
    idx = _oracle_index(oracle, ctx.prg_strings, w, k)
    assert int(idx["knode_base"][-1]) == ctx.n_knodes and len(idx["keys"]) == ctx.n_keys
    ocov, oprg, ocnt = _oracle_map(oracle, idx, bases, offsets, w, k, illumina, min_cluster_size, threads)
    ctx.reset()
    ctx.map_host(bases, offsets)
    gcov, gprg = ctx.coverage()
    gcnt = ctx.counters()
    if kernel != 2:  # the filtered kernel only counts the minimizers that are index keys
        assert gcnt["minimizers"] == ocnt["minimizers"]
    assert gcnt["hits"] == ocnt["hits"]
    assert gcnt["clusters_kept"] == ocnt["clusters_kept"]
    assert gcnt["hits_kept"] == ocnt["hits_kept"]
    assert np.array_equal(gprg, oprg)
    assert np.array_equal(gcov, ocov)
    # ... and the same batch in the 2-bit packed form (include/drprg_hip.h "packed reads"): identical vectors and counters, whatever
    # kernel sequence the context runs (the filtered sequence reads the words, the direct ones an expansion made on the device)
    from drprg_amd.pandora import pack_reads
    words, npos = pack_reads(bases)
    ctx.reset()
    ctx.map_host_packed(words, offsets, npos)
    pcov, pprg = ctx.coverage()
    pcnt = ctx.counters()
    assert np.array_equal(pcov, ocov) and np.array_equal(pprg, oprg)
    # (not leftover_reads: a non-ACGT byte reads as a letter in the words, so the packed filter can pass a position the ASCII filter
    # does not; verify_count_kernel rejects it, but it holds a slot of the ordered candidate list, the chunks of read_cluster_kernel shift
    # by it, and WHICH reads straddle a chunk's look-ahead and go through the generic pipeline instead can differ -- the results cannot)
    for key in ("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept"):
        assert pcnt[key] == gcnt[key], key
    # ... and, with the library of `make EXPERIMENTAL=1`, through the wave form of the last filtered stage (read_cluster_wave.hip; the
    # switch is read at every launch).  The default suite maps every case in the two input formats only.
    if EXPERIMENTAL and len(offsets) > 1 and int(offsets[-1]) // (len(offsets) - 1) <= 600:
        os.environ["DRPRG_RC_FORM"] = "wave"
        try:
            ctx.reset()
            ctx.map_host(bases, offsets)
            wcov, wprg = ctx.coverage()
            wcnt = ctx.counters()
        finally:
            del os.environ["DRPRG_RC_FORM"]
        assert np.array_equal(wcov, ocov) and np.array_equal(wprg, oprg)
        for key in ("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept"):  # (which reads are left to the generic
            assert wcnt[key] == gcnt[key], key                                               # pipeline is the form's own business)
    return ocnt


def _vcf_equals_oracle(ctx, oracle, tmp_path, panel, total_bases, illumina, genome_size, w=11, k=15):
    """pandora_genotyped.vcf of the coverage the DEVICE accumulated == the oracle's own VCF of that vector, byte for byte minus
    ##fileDate (tests/util.py oracle_vcf_text: oracle_params.c + oracle_vcf.c + oracle.c)"""
    from util import oracle_vcf_text, vcf_without_date
    vcf = str(tmp_path / "pandora_genotyped.vcf")
    info = ctx.genotype(str(tmp_path / "genes.fa"), vcf)
    covg, prg_reads = ctx.coverage()
    md, er = map_params(k, illumina)
    want, oinfo = oracle_vcf_text(oracle, panel.names, panel.prgs, dict(zip(panel.names, panel.refs)), covg, prg_reads, total_bases, w, k,
                                  genome_size, er)
    assert oinfo["e"] == info["exp_depth_covg"] and len(oinfo["present"]) == info["loci_present"]
    assert vcf_without_date(vcf) == want
    return info


CASES = [(11, 15, 1), (11, 15, 2), (11, 15, 3), (14, 15, 1), (14, 15, 2), (14, 15, 3), (5, 9, 1), (5, 9, 2), (5, 9, 3), (16, 13, 2),
         (16, 13, 3), (19, 21, 1), (19, 21, 3), (1, 15, 1), (1, 15, 2), (1, 15, 3), (11, 31, 1), (11, 31, 3)]


@pytest.mark.parametrize("w,k,kernel", CASES)
def test_short_reads_bit_exact(tmp_path, oracle, w, k, kernel):
    from drprg_amd import synth
    panel = synth.small_panel(seed=w * 100 + k)
    ctx = _ctx(tmp_path, panel, w, k, True, kernel=kernel)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 20000, seed=5)
    cnt = _compare(ctx, oracle, bases, offs, w, k, True, kernel)
    assert cnt["clusters_kept"] > 0
    if k >= 13 and w > 1 and not FORCED_GENERIC:  # (a 9-mer index matches everywhere: dozens of one-hit clusters per read; with w = 1
        # every k-mer is a minimizer: a read inside a locus carries 136 candidates, more than the look-ahead of a short-read chunk holds)
        assert ctx.counters()["leftover_reads"] == 0  # ordinary short reads never need the generic pipeline


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_long_reads_bit_exact(tmp_path, oracle, kernel):
    from drprg_amd import synth
    panel = synth.small_panel(seed=11, n_loci=6, length=1500)
    ctx = _ctx(tmp_path, panel, 11, 15, False, kernel=kernel)
    gen = synth.HaplotypeGenomes(panel, genome_size=60000, n_hap=4, seed=3)
    bases, offs = synth.sample_long_reads(gen, 1500, seed=3)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, False, kernel)
    assert cnt["clusters_kept"] > 0


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_ragged_and_degenerate_inputs(tmp_path, oracle, kernel):
    """empty reads, reads shorter than k, N runs, lower case, read boundaries at tile edges"""
    from drprg_amd import synth
    panel = synth.small_panel(seed=2)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=kernel)
    rng = np.random.default_rng(0)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=2, seed=3)
    g = gen.haps[0]
    reads = []
    for i in range(12000):
        L = int(rng.choice([0, 1, 14, 15, 24, 25, 26, 40, 150, 151, 300, 4064, 4096, 5000, 8160, 8192]))
        s = int(rng.integers(0, len(g) - L))
        r = g[s:s + L].copy()
        if L and rng.random() < 0.3:
            r[rng.integers(0, L, size=max(1, L // 50))] = ord("N")
        if L and rng.random() < 0.2:
            r = np.frombuffer(r.tobytes().lower(), dtype=np.uint8)
        reads.append(r)
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.concatenate(reads)
    _compare(ctx, oracle, bases, offs, 11, 15, True, kernel)
    # empty batch and a batch of only empty reads
    ctx.reset()
    ctx.map_host(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    ctx.map_host(np.zeros(0, np.uint8), np.zeros(5, np.uint64))
    cov, _ = ctx.coverage()
    assert cov.sum() == 0


def test_dense_panel_reads(tmp_path, oracle):
    """amplicon-like input: every read lies in the panel, so almost every tile position is a Bloom candidate"""
    from drprg_amd import synth
    panel = synth.small_panel(seed=6, n_loci=3, length=900)
    rng = np.random.default_rng(1)
    reads = []
    for i in range(6000):
        hap = np.frombuffer(synth.sample_haplotype(rng, panel.trees[i % 3]).encode(), np.uint8)
        s = int(rng.integers(0, len(hap) - 150))
        reads.append(hap[s:s + 150])
    offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)
    bases = np.concatenate(reads)
    for kernel in (1, 2, 3):
        ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=kernel)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, kernel)
        assert cnt["clusters_kept"] > 5000


_RC = bytes.maketrans(b"ACGT", b"TGCA")


def _reads_from(rng, seqs, n, length, sub_rate=0.002):
    """n reads of `length` bases drawn from the given sequences (random strand, a few substitutions)"""
    reads = []
    for i in range(n):
        hap = seqs[i % len(seqs)]
        ln = min(length, len(hap))
        s = int(rng.integers(0, len(hap) - ln + 1))
        r = hap[s:s + ln]
        if rng.random() < 0.5:
            r = r.translate(_RC)[::-1]
        a = np.frombuffer(r, np.uint8).copy()
        err = np.nonzero(rng.random(ln) < sub_rate)[0]
        a[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=err.size)]
        reads.append(a)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return np.concatenate(reads), offs


@pytest.mark.experimental
@pytest.mark.parametrize("w", [11, 14])
def test_read_by_read_verification(tmp_path, oracle, w):
    """read_verify_kernel (short-read batches at k = 15: every read with several candidates is sketched once) on everything its
    chunk logic distinguishes: 150-base panel reads (sketched), genome reads with a stray index k-mer (the lane path's queue), panel
    reads of 280 / 900 bases (more candidates than a chunk's look-ahead: the overhang goes through the queue of another chunk),
    reads over 1024 bases (never sketched), reads shorter than k, empty reads, N runs, lower case -- mean length below 300, the batches
    the kernel serves when DRPRG_VERIFY_FORM=read asks for it; and the default lane form must count the same"""
    from drprg_amd import synth
    rng = np.random.default_rng(40 + w)
    loci = [synth.make_locus(rng, 2500, site_every=45) for _ in range(3)]
    panel = synth.Panel(["a", "b", "c"], loci)
    haps = [synth.sample_haplotype(rng, t).encode() for t in loci for _ in range(3)]
    genome = synth.random_seq(rng, 60000).encode()
    parts = [_reads_from(rng, haps, 5000, 150), _reads_from(rng, [genome], 9000, 150), _reads_from(rng, haps, 600, 280),
             _reads_from(rng, haps, 60, 900, sub_rate=0.01), _reads_from(rng, haps, 12, 1500), _reads_from(rng, haps + [genome], 800, 40),
             _reads_from(rng, haps, 300, 14), _reads_from(rng, haps, 300, 15), _reads_from(rng, haps, 300, 16)]
    reads = []
    for b, o in parts:
        reads += [b[int(o[i]):int(o[i + 1])].copy() for i in range(len(o) - 1)]
    reads += [np.zeros(0, np.uint8)] * 200
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    for r in reads:
        if len(r) and rng.random() < 0.05:
            r[rng.integers(0, len(r), size=max(1, len(r) // 60))] = ord("N")
    reads = [np.frombuffer(r.tobytes().lower(), np.uint8) if len(r) and rng.random() < 0.1 else r for r in reads]
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.concatenate(reads)
    assert int(offs[-1]) // len(reads) <= 300
    ctx = _ctx(tmp_path, panel, w, 15, True, genome_size=60000, kernel=2)
    os.environ["DRPRG_VERIFY_FORM"] = "read"
    try:
        cnt = _compare(ctx, oracle, bases, offs, w, 15, True, 2)
        assert cnt["clusters_kept"] > 4000
        ctx.reset()
        ctx.map_host(bases, offs)
        got = ctx.counters()
        cov, prg = ctx.coverage()
    finally:
        del os.environ["DRPRG_VERIFY_FORM"]
    ctx.reset()
    ctx.map_host(bases, offs)
    lane = ctx.counters()
    lcov, lprg = ctx.coverage()
    assert np.array_equal(cov, lcov) and np.array_equal(prg, lprg)
    for key in ("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept", "leftover_reads"):
        assert lane[key] == got[key], key


def test_candidates_at_both_ends_of_the_batch(tmp_path, oracle):
    """verify_scan_kernel reads the candidates from the filter kernel's slices through the prefix of the superblock counts it keeps
    in LDS: a batch whose candidates sit in its first and its last slices, with half a million reads that leave no candidate between
    them, makes the workgroup whose share straddles the gap bisect across thousands of empty superblocks -- and the three-kernel
    sequence (DRPRG_VERIFY_FORM=gather) must count the same"""
    from drprg_amd import synth
    rng = np.random.default_rng(77)
    panel = synth.small_panel(seed=31, n_loci=3, length=900)
    haps = [synth.sample_haplotype(rng, t).encode() for t in panel.trees]
    head, ho = _reads_from(rng, haps, 180, 150)
    tail, to = _reads_from(rng, haps, 120, 150)
    n_mid = 500_000
    mid = np.full(n_mid * 150, ord("A"), np.uint8)
    bases = np.concatenate([head, mid, tail])
    offs = np.concatenate([ho, ho[-1] + np.arange(1, n_mid + 1, dtype=np.uint64) * np.uint64(150), ho[-1] + np.uint64(n_mid * 150) + to[1:]])
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 2, threads=ORACLE_THREADS)
    assert cnt["clusters_kept"] > 200
    got = ctx.counters()
    os.environ["DRPRG_VERIFY_FORM"] = "gather"
    try:
        ctx.reset()
        ctx.map_host(bases, offs)
        old = ctx.counters()
        ocov, oprg = ctx.coverage()
    finally:
        del os.environ["DRPRG_VERIFY_FORM"]
    ctx.reset()
    ctx.map_host(bases, offs)
    cov, prg = ctx.coverage()
    assert np.array_equal(cov, ocov) and np.array_equal(prg, oprg)
    for key in ("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept", "leftover_reads"):
        assert old[key] == got[key] or key == "leftover_reads", key


@pytest.mark.parametrize("illumina", [True, False])
def test_reads_with_hits_in_several_groups(tmp_path, oracle, illumina):
    """duplicated loci, a reverse-complemented copy and an inverted repeat: every read has hits in several
    (prg, strand) groups, so read_cluster_kernel takes its wave path and the overlap sweep decides (equal clusters on two
    PRGs, same PRG on both strands); both kernels against the oracle"""
    from drprg_amd import synth
    rng = np.random.default_rng(5)
    a = synth.make_locus(rng, 800, site_every=50)
    b = synth.make_locus(rng, 600, site_every=50)
    ref_a = synth.sample_haplotype(None, a, first_allele=True)
    x = synth.random_seq(rng, 300)
    inverted = x + synth.random_seq(rng, 40) + x.encode().translate(_RC)[::-1].decode() + synth.random_seq(rng, 60)
    panel = synth.Panel(["a", "a_copy", "a_rc", "b", "inv"],
                        [a, a, [ref_a.encode().translate(_RC)[::-1].decode()], b, [inverted]])
    seqs = [synth.sample_haplotype(rng, t).encode() for t in (a, a, b) for _ in range(3)] + [inverted.encode()]
    bases, offs = _reads_from(rng, seqs, 4000, 150)
    lb, lo = _reads_from(rng, seqs, 300, 700)  # several clusters per read with the Illumina gap limit
    bases = np.concatenate([bases, lb])
    offs = np.concatenate([offs, lo[1:] + offs[-1]])
    nb, no = _reads_from(rng, seqs, 300, 700, sub_rate=0.06)  # errors tear the hit runs: many clusters per group, hundreds of hits
    bases = np.concatenate([bases, nb])
    offs = np.concatenate([offs, no[1:] + offs[-1]])
    for kernel in (1, 2, 3):
        ctx = _ctx(tmp_path, panel, 11, 15, illumina, kernel=kernel)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, illumina, kernel)
        assert cnt["clusters_kept"] > 1000


@pytest.mark.parametrize("copies", [70, 160])
def test_reads_that_do_not_fit_the_per_read_kernel(tmp_path, oracle, copies):
    """a k-mer shared by 70 PRGs gives a read 70 clusters (more than the 64 lanes of the wave path), 160 copies more hits
    than a chunk stages: such reads are left to the generic pipeline, next to ordinary reads in the same batch"""
    from drprg_amd import synth
    rng = np.random.default_rng(9)
    rep = synth.make_locus(rng, 400, site_every=70)
    single = synth.make_locus(rng, 900, site_every=50)
    panel = synth.Panel([f"rep{i}" for i in range(copies)] + ["single"], [rep] * copies + [single])
    seqs = [synth.sample_haplotype(rng, rep).encode(), synth.sample_haplotype(rng, single).encode(),
            synth.random_seq(rng, 5000).encode()]
    bases, offs = _reads_from(rng, seqs, 1500, 150)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 2)
    assert cnt["clusters_kept"] > 500
    if not FORCED_GENERIC:
        assert 400 < ctx.counters()["leftover_reads"] < 1100  # the reads of the repeated locus, not the others


def test_reads_longer_than_the_staged_range(tmp_path, oracle):
    """long reads that lie entirely inside a long locus have thousands of minimizer hits, more than read_cluster_kernel
    stages for one read: they take the generic pipeline while the short reads of the same batch do not"""
    from drprg_amd import synth
    rng = np.random.default_rng(12)
    long_locus = synth.make_locus(rng, 16000, site_every=80)
    short_locus = synth.make_locus(rng, 800, site_every=50)
    panel = synth.Panel(["long", "short"], [long_locus, short_locus])
    seqs_long = [synth.sample_haplotype(rng, long_locus).encode() for _ in range(3)]
    seqs_short = [synth.sample_haplotype(rng, short_locus).encode() for _ in range(3)]
    b1, o1 = _reads_from(rng, seqs_long, 60, 9000, sub_rate=0.01)
    b2, o2 = _reads_from(rng, seqs_short + seqs_long, 1500, 150)
    b3, o3 = _reads_from(rng, seqs_long, 40, 3000, sub_rate=0.03)
    bases = np.concatenate([b1, b2, b3])
    offs = np.concatenate([o1, o2[1:] + o1[-1], o3[1:] + o1[-1] + o2[-1]])
    for kernel in (1, 2, 3):
        ctx = _ctx(tmp_path, panel, 11, 15, False, kernel=kernel)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, False, kernel)
        assert cnt["clusters_kept"] > 500
        if kernel == 2 and not FORCED_GENERIC:
            assert 20 <= ctx.counters()["leftover_reads"] <= 100  # most 9 kb reads (a few fit: 2048 candidates are staged)


def test_many_small_prgs(tmp_path, oracle):
    """3000 loci of 56 bp (the most the LDS filter takes at ~8 k-mer nodes each): PRG ids up to 12 bits in the hit key, a
    12 KB per-PRG histogram in read_cluster_kernel and cluster_count_kernel, every read spans three loci (three clusters
    of a few hits each: min_cluster_size 2)"""
    from drprg_amd import synth
    rng = np.random.default_rng(17)
    loci = [[synth.random_seq(rng, 56)] for _ in range(3000)]
    panel = synth.Panel([f"p{i}" for i in range(3000)], loci)
    genome = "".join(l[0] for l in loci).encode()
    bases, offs = _reads_from(rng, [genome], 6000, 150, sub_rate=0.001)
    for kernel in (1, 2, 3):
        ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=kernel, min_cluster_size=2)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, kernel, min_cluster_size=2)
        assert cnt["clusters_kept"] > 1000


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRPRG_FUZZ_SEEDS", "12"))))
def test_randomized_configurations(tmp_path, oracle, seed):
    """seeded random (w, k, technology, min_cluster_size, panel shape, read-length mix): ragged short reads, reads across
    duplicated loci, a few long ones -- whatever combination of the per-read kernel's paths that produces"""
    from drprg_amd import synth
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([9, 11, 13, 15, 15, 15]))
    w = int(rng.integers(1, 17))
    illumina = bool(rng.integers(0, 2))
    mcs = int(rng.choice([1, 2, 5, 10]))
    n_loci = int(rng.integers(2, 7))
    trees = [synth.make_locus(rng, int(rng.integers(300, 1500)), site_every=int(rng.integers(25, 90))) for _ in range(n_loci)]
    if rng.random() < 0.5:
        trees.append(trees[0])  # a duplicated locus: hits in two groups
    panel = synth.Panel([f"l{i}" for i in range(len(trees))], trees)
    seqs = [synth.sample_haplotype(rng, t).encode() for t in trees for _ in range(2)] + [synth.random_seq(rng, 4000).encode()]
    parts = []
    for length, n in ((int(rng.integers(16, 60)), 300), (150, 1500), (int(rng.integers(200, 700)), 400), (int(rng.integers(1500, 4000)), 40)):
        parts.append(_reads_from(rng, seqs, n, length, sub_rate=float(rng.choice([0.0, 0.002, 0.02]))))
    bases = np.concatenate([p[0] for p in parts])
    offs = [np.zeros(1, np.uint64)]
    for p in parts:
        offs.append(p[1][1:] + offs[-1][-1])
    offs = np.concatenate(offs)
    for kernel in (1, 2, 3):
        ctx = _ctx(tmp_path, panel, w, k, illumina, kernel=kernel, min_cluster_size=mcs)
        _compare(ctx, oracle, bases, offs, w, k, illumina, kernel, min_cluster_size=mcs)
    if k == 15:  # the same batch through the middle tier of the filter (tables built for this small index on request)
        os.environ["DRPRG_FORCE_MID_TIER"] = "1"
        try:
            ctx = _ctx(tmp_path, panel, w, k, illumina, kernel=2, min_cluster_size=mcs)
            assert ctx.table_tier()["l2_filter_bytes"] > 0
            _compare(ctx, oracle, bases, offs, w, k, illumina, 2, min_cluster_size=mcs)
        finally:
            del os.environ["DRPRG_FORCE_MID_TIER"]


def test_batches_accumulate(tmp_path, oracle):
    """mapping in several batches == mapping in one (coverage is additive)"""
    from drprg_amd import synth
    panel = synth.small_panel(seed=4)
    ctx = _ctx(tmp_path, panel, 11, 15, True)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 9000, seed=8)
    ctx.map_host(bases, offs)
    one, _ = ctx.coverage()
    ctx.reset()
    for lo in range(0, 9000, 3000):
        b = bases[int(offs[lo]):int(offs[lo + 3000])]
        o = offs[lo:lo + 3001] - offs[lo]
        ctx.map_host(b, o)
    three, _ = ctx.coverage()
    assert np.array_equal(one, three)


def test_filter_kernel_refuses_unsupported_parameters(tmp_path):
    from drprg_amd import DependencyError, synth
    panel = synth.small_panel(seed=4)
    with pytest.raises(DependencyError):
        _ctx(tmp_path, panel, 11, 21, True, kernel=2)


def test_cli_map_end_to_end(tmp_path, oracle):
    """configs[0]-style plumbing: the drop-in `pandora` executable with the argv drprg builds
    (/root/reference/src/lib.rs:594-618), FASTQ(.gz) in, pandora_genotyped.vcf out; same VCF as the in-process path
    fed with the oracle's coverage."""
    import os
    import subprocess
    from drprg_amd import Context, synth
    from drprg_amd._lib import PANDORA_EXE
    w, k = 11, 15
    panel = synth.small_panel(seed=33)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    gen = synth.HaplotypeGenomes(panel, genome_size=4411532 // 100, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 10000, seed=1)
    fq = str(tmp_path / "reads.fq.gz")
    synth.write_fastq(fq, bases, offs, gz=True)
    r = subprocess.run([PANDORA_EXE, "index", "-t", "2", "-w", str(w), "-k", str(k), prg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "out"
    argv = [PANDORA_EXE, "map", "--genotype", "--local", "--gt-conf", "0", "-v", "-o", str(out), "-g", "4411532", "--max-covg",
            "4294967295", "--vcf-refs", genes, "-t", "1", "-w", str(w), "-k", str(k), "-c", "10", "-I", prg, fq]
    r = subprocess.run(argv, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    vcf = out / "pandora_genotyped.vcf"
    assert vcf.exists()
    # discover (-o <out>/discover, /root/reference/src/predict.rs:248): a parsable denovo_paths.txt with zero loci
    # (/root/reference/src/lib.rs:648-697), candidate regions, a loud warning; the `map` that follows in the same <out> takes
    # the coverage vector discover left there instead of mapping the reads a second time, and writes the identical VCF
    q = tmp_path / "query.tsv"
    q.write_text(f"sample\t{fq}\n")
    out2 = tmp_path / "out2"
    r = subprocess.run([PANDORA_EXE, "discover", "-g", "4411532", "--max-covg", "4294967295", "-v", "-o", str(out2 / "discover"),
                        "-t", "1", "-w", str(w), "-k", str(k), "-c", "10", "-I", prg, str(q)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "discover:" in r.stdout and (out2 / "discover" / "denovo_variants.tsv").exists()
    from drprg_amd import Pandora
    assert Pandora.list_prgs_with_novel_variants(str(out2 / "discover" / "denovo_paths.txt")) == []
    assert (out2 / "discover" / "candidate_regions.tsv").exists()
    argv2 = list(argv)
    argv2[argv2.index("-o") + 1] = str(out2)
    r = subprocess.run(argv2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "no second mapping pass" in r.stdout
    keep = lambda p: [l for l in open(p) if not l.startswith("##fileDate")]
    assert keep(out2 / "pandora_genotyped.vcf") == keep(vcf)
    # other parameters (here -c 11): the cached vector is not this run's, the reads are mapped
    argv3 = list(argv2)
    argv3[argv3.index("-c") + 1] = "11"
    r = subprocess.run(argv3, capture_output=True, text=True)
    assert r.returncode == 0 and "no second mapping pass" not in r.stdout and "reads=10000" in r.stdout
    # reference VCF: host genotyper on the oracle's coverage of the same reads (oracle's own index of the same PRGs)
    ctx = Context(prg, w, k, device=-1, from_files=True)
    ctx.set_opts(illumina=True, genome_size=4411532)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    ref = str(tmp_path / "ref.vcf")
    ctx.genotype(genes, ref)
    strip = lambda p: [l for l in open(p) if not l.startswith("##fileDate")]
    assert strip(vcf) == strip(ref)


# ---- the BASELINE.json configurations ----------------------------------------------------------------------------
# configs[1] 10M x 150 bp vs the mtb index, configs[2] Nanopore reads vs the mtb index, configs[4] the 500-locus /
# 50k-variant index.  Two mtb-like indexes: synth.mtb_like_panel() (random backbone, the bench's alternative workload) and
# the SURVEY 8d index (backbone = the reference's genes.fa, sites = its panel.bcf records + seeded bubbles; bench.py's headline).  Oracle parity on read counts the
# oracle maps in seconds on the host's cores; the full BASELINE sizes through size-independent properties.
ORACLE_THREADS = max(1, min(os.cpu_count() or 1, 32))
GOLDEN_INDEX_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "downstream")
_PANELS = {}


def _baseline_panel(name):
    from drprg_amd import synth
    if name not in _PANELS:
        if name == "mtb_like":
            _PANELS[name] = synth.mtb_like_panel()
        elif name == "mtb_8d":
            _PANELS[name] = synth.mtb_8d_panel(GOLDEN_INDEX_DIR)
        else:
            _PANELS[name] = synth.big_panel()
        _PANELS[name + "/genomes"] = synth.HaplotypeGenomes(_PANELS[name], n_hap=8)
    return _PANELS[name], _PANELS[name + "/genomes"]


def _baseline_reads(name, long_reads, n_reads):
    """seeded reads of a BASELINE configuration (sampled once per session: the three kernel sequences map the same batch)"""
    from drprg_amd import synth
    key = f"{name}/reads/{int(long_reads)}"
    if key not in _PANELS:
        genomes = _baseline_panel(name)[1]
        _PANELS[key] = synth.sample_long_reads(genomes, n_reads, seed=3) if long_reads else synth.sample_short_reads(genomes, n_reads, seed=2)
    return _PANELS[key]


def _expected_kernel(name, kernel):
    """what `auto` resolves to: the LDS filter for the mtb-sized indexes, the candidate form for the 500-locus index"""
    return kernel or (3 if name == "big" else 2)


@pytest.mark.parametrize("kernel", [0, 1, 3])
@pytest.mark.parametrize("name", ["mtb_like", "mtb_8d"])
def test_config1_illumina_reads_vs_mtb_index(tmp_path, oracle, name, kernel):
    """configs[1] at 1.2 M reads, 4.4 Mb background genome: auto (= the Bloom-prefiltered sequence), the direct kernel with
    the generic cluster pipeline, and the direct kernel's candidate form, all against the oracle"""
    from drprg_amd import synth
    panel, genomes = _baseline_panel(name)
    ctx = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE, kernel=kernel)
    bases, offs = _baseline_reads(name, False, 1_200_000)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, kernel or 2, threads=ORACLE_THREADS)
    assert ctx.counters()["kernel"] == _expected_kernel(name, kernel)
    assert cnt["clusters_kept"] > 5000
    if not FORCED_GENERIC:
        assert ctx.counters()["leftover_reads"] == 0
    if kernel == 0:
        info = _vcf_equals_oracle(ctx, oracle, tmp_path, panel, int(offs[-1]), True, synth.MTB_GENOME_SIZE)
        assert info["records"] > 200 and info["loci_present"] == 18


@pytest.mark.parametrize("kernel", [0, 1, 3])
@pytest.mark.parametrize("name", ["mtb_like", "mtb_8d"])
def test_config2_nanopore_reads_vs_mtb_index(tmp_path, oracle, name, kernel):
    """configs[2] at 24 k reads (mean 4 kb, 5 % errors split 40/30/30 sub/ins/del), max_diff 250, error rate 0.11"""
    from drprg_amd import synth
    panel, genomes = _baseline_panel(name)
    ctx = _ctx(tmp_path, panel, 11, 15, False, genome_size=synth.MTB_GENOME_SIZE, kernel=kernel)
    bases, offs = _baseline_reads(name, True, 24_000)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, False, kernel or 2, threads=ORACLE_THREADS)
    assert ctx.counters()["kernel"] == _expected_kernel(name, kernel)
    assert cnt["clusters_kept"] > 200


@pytest.mark.parametrize("kernel", [0, 1])
def test_config4_500_locus_index(tmp_path, oracle, kernel):
    """configs[4]: 500 loci / 50 k sites / 573 k keys.  No LDS filter at this size: `auto` is the direct kernel in its
    candidate form behind the L2-resident Bloom tier and a 12 MB probe table; kernel 1 the same kernel + generic pipeline."""
    from drprg_amd import DependencyError, synth
    panel, genomes = _baseline_panel("big")
    ctx = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE, kernel=kernel)
    assert ctx.n_keys > 500_000 and ctx.n_prgs == 500
    assert ctx.table_tier()["lds_filter_bytes"] == 0 and ctx.table_tier()["l2_filter_bytes"] == 0  # no filter tier serves this index ...
    with pytest.raises(DependencyError):
        ctx.set_opts(illumina=True, genome_size=synth.MTB_GENOME_SIZE, kernel=2)  # ... so the filtered sequence is refused
    ctx.set_opts(illumina=True, genome_size=synth.MTB_GENOME_SIZE, kernel=kernel)
    assert ctx.table_tier()["pbloom_words"] > 0 and ctx.table_tier()["table_bytes"] > 8 << 20
    bases, offs = _baseline_reads("big", False, 600_000)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, kernel or 3, threads=ORACLE_THREADS)
    assert ctx.counters()["kernel"] == _expected_kernel("big", kernel)
    assert cnt["clusters_kept"] > 50_000
    if kernel == 0:
        info = _vcf_equals_oracle(ctx, oracle, tmp_path, panel, int(offs[-1]), True, synth.MTB_GENOME_SIZE)
        assert info["records"] > 20_000 and info["loci_present"] == 500


def _device_reads(torch, genomes, n_reads, seed, long_reads=False):
    import bench
    dev = torch.device("cuda", 0)
    hap_pad = torch.from_numpy(genomes.padded()).to(dev)
    hap_lens = torch.from_numpy(genomes.lens).to(dev)
    if long_reads:
        return bench.gpu_sample_long_reads(torch, hap_pad, hap_lens, n_reads, seed, dev)
    return bench.gpu_sample_reads(torch, hap_pad, hap_lens, n_reads, 150, seed, dev)


@pytest.mark.parametrize("name,illumina,n_reads", [("mtb_8d", True, 10_000_000), ("mtb_8d", False, 2_000_000), ("big", True, 10_000_000)],
                         ids=["config1_10M_illumina", "config2_2M_nanopore", "config4_10M_500_loci"])
def test_full_size_properties(tmp_path, name, illumina, n_reads):
    """BASELINE.json's full sizes, where the oracle cannot follow in seconds: (1) sharding invariance -- coverage(whole batch)
    == coverage(first half) + coverage(second half), bit for bit (what makes the multi-GPU reduce exact); (2) the sequence
    `auto` picks and the direct kernel + generic cluster pipeline (radix sort, cluster kernels) give the identical vector;
    (3) a second pass over the same batch doubles every counter (accumulation, no lost or duplicated atomics)."""
    import torch
    import bench
    from drprg_amd import synth
    panel, genomes = _baseline_panel(name)
    ctx = _ctx(tmp_path, panel, 11, 15, illumina, genome_size=synth.MTB_GENOME_SIZE)
    bases, offsets = _device_reads(torch, genomes, n_reads, 2, long_reads=not illumina)
    dev = bases.device
    stream = torch.cuda.Stream(dev)
    covg = torch.zeros(2 * ctx.n_knodes, dtype=torch.int32, device=dev)
    prg_reads = torch.zeros(ctx.n_prgs, dtype=torch.int32, device=dev)
    bench.map_range(ctx, bases, offsets, 0, n_reads, covg, prg_reads, stream, torch)
    assert int(covg.sum().item()) > 100_000
    opts = dict(illumina=illumina, min_cluster_size=10, genome_size=synth.MTB_GENOME_SIZE)
    shard_invariant, kernels_agree = bench.full_size_checks(torch, ctx, opts, bases, offsets, n_reads, covg, prg_reads, stream)
    assert shard_invariant is True and kernels_agree is True
    twice = covg.clone()
    bench.map_range(ctx, bases, offsets, 0, n_reads, twice, prg_reads, stream, torch)
    assert torch.equal(twice, 2 * covg)


def test_multi_device_context_equals_single(tmp_path, oracle):
    """drprg_hip_open_multi (native host path): map_fastx shards the ingest blocks over the listed devices and sums
    their coverage vectors into the first.  On a one-GPU box the list names device 0 twice -- two mappers, two streams, the
    same code path as two GPUs -- and the result must equal the single-device context and the oracle, counters included;
    the CLI takes the list from DRPRG_HIP_DEVICES."""
    import subprocess
    from drprg_amd import Context, synth
    from drprg_amd._lib import PANDORA_EXE
    w, k = 11, 15
    panel = synth.small_panel(seed=17, n_loci=5, length=900)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    gen = synth.HaplotypeGenomes(panel, genome_size=60000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 1_500_000, seed=4)  # ~10 ingest blocks
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq_fixed(fq, bases, 150)
    single = Context(prg, w, k, device=0, from_files=False)
    single.set_opts(illumina=True, genome_size=60000)
    single.set_threads(8)
    single.map_fastx(fq)
    want, want_prg = single.coverage()
    multi = Context(prg, w, k, from_files=False, devices=[0, 0, 0])
    multi.set_opts(illumina=True, genome_size=60000)
    multi.set_threads(8)
    multi.map_fastx(fq)
    got, got_prg = multi.coverage()
    assert np.array_equal(got, want) and np.array_equal(got_prg, want_prg)
    # the vectors were summed on the device (one GPU listed three times cannot form an RCCL communicator: peer copy + add kernel)
    assert multi.reduce().startswith("device add") or multi.reduce().startswith("rccl")
    assert np.array_equal(multi.coverage()[0], want)  # (a second reduce finds the other devices empty)
    cs, cm = single.counters(), multi.counters()
    for key in ("reads", "bases", "hits", "clusters_kept", "hits_kept"):
        assert cs[key] == cm[key], key
    # ... and again with contexts that have never mapped anything (first-call allocations while the other mappers of the device are
    # busy: a fresh candidate array used to be zeroed by a null-stream memset that could land after the first batch had written it --
    # one such run in ten lost clusters; tools/stress_multi.py runs this hundreds of times)
    for _ in range(12):
        cold = Context(prg, w, k, from_files=False, devices=[0, 0, 0])
        cold.set_opts(illumina=True, genome_size=60000)
        cold.set_threads(8)
        cold.map_fastx(fq)
        assert np.array_equal(cold.coverage()[0], want)
        cold.close()
    idx = oracle.build_index(panel.prgs, w, k)
    ocov, oprg, _ = _oracle_map(oracle, idx, bases, offs, w, k, True, threads=ORACLE_THREADS)
    assert np.array_equal(got, ocov) and np.array_equal(got_prg, oprg)
    # a second file accumulates on top, and reset clears every device
    multi.map_fastx(fq)
    assert np.array_equal(multi.coverage()[0], 2 * want)
    multi.reset()
    assert multi.coverage()[0].sum() == 0 and multi.counters()["reads"] == 0
    # the drop-in executable over "two devices"
    r = subprocess.run([PANDORA_EXE, "index", "-t", "2", "-w", str(w), "-k", str(k), prg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    outs = []
    for devs in ("0", "0,0"):
        out = tmp_path / f"out_{devs.replace(',', '_')}"
        r = subprocess.run([PANDORA_EXE, "map", "--genotype", "--local", "-o", str(out), "-g", "60000", "--vcf-refs", genes, "-t", "8", "-w", str(w),
                            "-k", str(k), "-c", "10", "-I", prg, fq], capture_output=True, text=True, env=dict(os.environ, DRPRG_HIP_DEVICES=devs))
        assert r.returncode == 0, r.stderr
        outs.append([l for l in open(out / "pandora_genotyped.vcf") if not l.startswith("##fileDate")])
    assert outs[0] == outs[1] and len(outs[0]) > 30


@pytest.mark.experimental
def test_in_kernel_clustering_of_sketch_wave_kernel(tmp_path, oracle, monkeypatch):
    """DRPRG_WAVE_FUSE=1 (opt-in): sketch_wave_kernel clusters the reads that lie inside one of its tiles itself -- single
    (prg, strand) group, minimizers with up to eight index records -- and only the others (reads across a tile edge, hits in
    several groups, more than 64 index minimizers) leave records for gather + read_cluster_kernel.  Same coverage vector and
    counters as the oracle on the 500-locus index and on a nested / indel panel with duplicated loci; a tiny record slice
    forces the overflow + undo + regrow path."""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_WAVE_FUSE", "1")
    panel, genomes = _baseline_panel("big")
    bases, offs = _baseline_reads("big", False, 600_000)
    ctx = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE, kernel=3)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 3, threads=ORACLE_THREADS)
    assert cnt["clusters_kept"] > 50_000 and ctx.counters()["leftover_reads"] == 0
    ctx.close()
    rng = np.random.default_rng(23)
    a = synth.make_locus(rng, 900, site_every=35, nested_frac=0.3, indel_frac=0.2)
    b = synth.make_locus(rng, 700, site_every=35, nested_frac=0.3, indel_frac=0.2)
    small = synth.Panel(["a", "a_copy", "b"], [a, a, b])
    seqs = [synth.sample_haplotype(rng, t).encode() for t in (a, b) for _ in range(4)] + [synth.random_seq(rng, 3000).encode()]
    parts = [_reads_from(rng, seqs, 6000, 150), _reads_from(rng, seqs, 500, 400), _reads_from(rng, seqs, 60, 2500)]
    sb = np.concatenate([p[0] for p in parts])
    so = np.concatenate([np.zeros(1, np.uint64)] + [p[1][1:] + sum(int(q[1][-1]) for q in parts[:i]) for i, p in enumerate(parts)])
    for illumina in (True, False):
        ctx = _ctx(tmp_path, small, 11, 15, illumina, kernel=3)
        _compare(ctx, oracle, sb, so, 11, 15, illumina, 3)
        ctx.close()


@pytest.mark.parametrize("fuse", ["0", pytest.param("1", marks=pytest.mark.experimental)])
def test_wave_tile_geometry_edges(tmp_path, oracle, monkeypatch, fuse):
    """sketch_wave_kernel evaluates 61 lanes x 16 positions per tile (976), loads 1024 bases, and a read is clustered in-kernel
    (DRPRG_WAVE_FUSE=1) only if all its k-mers start inside one tile: reads whose lengths sit on those edges (15, 16, 17, 31,
    960 ... 1007, 1024, 1952), runs of empty reads (several reads starting at one position), N runs that cross lane and tile
    borders, all drawn from loci so that they carry hits"""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_WAVE_FUSE", fuse)
    rng = np.random.default_rng(41)
    loci = [synth.make_locus(rng, 2600, site_every=45, nested_frac=0.2, indel_frac=0.1) for _ in range(3)]
    panel = synth.Panel(["x", "y", "z"], loci)
    haps = [np.frombuffer(synth.sample_haplotype(rng, t).encode(), np.uint8) for t in loci for _ in range(2)]
    lengths = [0, 0, 0, 1, 14, 15, 16, 17, 30, 31, 32, 150, 150, 150, 959, 960, 961, 975, 976, 977, 990, 991, 992, 1005, 1006, 1007, 1008, 1023, 1024,
               1025, 1951, 1952, 1953]
    reads = []
    for i in range(9000):
        L = int(lengths[int(rng.integers(0, len(lengths)))])
        h = haps[i % len(haps)]
        s = int(rng.integers(0, len(h) - L + 1))
        r = h[s:s + L].copy()
        if L and rng.random() < 0.15:  # an N run of 1..40 bases somewhere
            a0 = int(rng.integers(0, L))
            r[a0:a0 + int(rng.integers(1, 41))] = ord("N")
        if L and rng.random() < 0.5:
            r = synth._COMP[r[::-1]]
        reads.append(r)
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.concatenate(reads)
    for illumina in (True, False):
        ctx = _ctx(tmp_path, panel, 11, 15, illumina, kernel=3)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, illumina, 3)
        assert cnt["clusters_kept"] > 1000
        ctx.close()
    ctx = _ctx(tmp_path, panel, 14, 15, True, kernel=3)  # the other instantiation of the kernel
    _compare(ctx, oracle, bases, offs, 14, 15, True, 3)


@pytest.mark.parametrize("kernel", [0, 1, 3])
def test_deferred_batches_equal_synchronous_ones(tmp_path, oracle, kernel):
    """drprg_hip_map_device_async: a batch is queued and its read-back is looked at while the next one runs.  Batches that need the
    host afterwards -- dense reads that overflow the candidate slices (run again with larger buffers), reads of a 70-copy locus
    that read_cluster_kernel leaves to the generic pipeline -- come out exactly as through the synchronous call and as the oracle
    has them; kernel=1 (no deferred form) falls back to the synchronous path behind the same entry."""
    import torch
    from drprg_amd import synth
    rng = np.random.default_rng(21)
    rep = synth.make_locus(rng, 400, site_every=70)
    single = synth.make_locus(rng, 900, site_every=50)
    other = synth.make_locus(rng, 1200, site_every=40)
    panel = synth.Panel([f"rep{i}" for i in range(70)] + ["single", "other"], [rep] * 70 + [single, other])
    hap = lambda t: synth.sample_haplotype(rng, t).encode()
    background = synth.random_seq(rng, 30000).encode()
    batches = [
        _reads_from(rng, [hap(single), hap(other)], 6000, 150),                       # dense: every read inside the panel
        _reads_from(rng, [hap(rep), hap(single), background], 1500, 150),             # leftovers of the repeated locus
        _reads_from(rng, [background, hap(other)], 20000, 150),                       # mostly off-panel
        _reads_from(rng, [hap(other), hap(single), hap(rep)], 9000, 150),             # dense again, larger than the first
        _reads_from(rng, [hap(other)], 10, 150),                                      # a tiny last batch
    ]
    dev = torch.device("cuda", 0)
    tens = [(torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), len(o) - 1, int(o[-1]))
            for b, o in batches]
    torch.cuda.synchronize()
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=kernel)
    for tb, to, n, nb in tens:
        ctx.map_device(tb.data_ptr(), to.data_ptr(), n, nb)
    want, want_prg = ctx.coverage()
    want_cnt = ctx.counters()
    ctx.reset()
    for tb, to, n, nb in tens:
        ctx.map_device_async(tb.data_ptr(), to.data_ptr(), n, nb)
    got, got_prg = ctx.coverage()  # (reading results completes the batch in flight)
    cnt = ctx.counters()
    assert np.array_equal(got, want) and np.array_equal(got_prg, want_prg)
    for key in ("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept", "leftover_reads"):
        assert cnt[key] == want_cnt[key], key
    if kernel == 0 and not FORCED_GENERIC:
        assert cnt["kernel"] == 2 and cnt["leftover_reads"] > 300
    # ... and as the oracle has them
    bases = np.concatenate([b for b, _ in batches])
    offs = np.concatenate([[0]] + [o[1:] + sum(int(p[-1]) for _, p in batches[:i]) for i, (_, o) in enumerate(batches)]).astype(np.uint64)
    idx = _oracle_index(oracle, ctx.prg_strings, 11, 15)
    ocov, oprg, _ = _oracle_map(oracle, idx, bases, offs, 11, 15, True)
    assert np.array_equal(got, ocov) and np.array_equal(got_prg, oprg)
    # a second round into caller-owned accumulators that alternate, with explicit sync
    ctx.reset()
    accs = [torch.zeros(2 * ctx.n_knodes + ctx.n_prgs, dtype=torch.int32, device=dev) for _ in range(2)]
    sums = np.zeros(2 * ctx.n_knodes, dtype=np.int64)
    for i, (tb, to, n, nb) in enumerate(tens):
        a = accs[i % 2]
        if i >= 2:
            sums += a[:2 * ctx.n_knodes].cpu().numpy()  # batch i-2 was completed by call i-1
        a.zero_()
        torch.cuda.synchronize()
        ctx.map_device_async(tb.data_ptr(), to.data_ptr(), n, nb, a.data_ptr(), a.data_ptr() + 8 * ctx.n_knodes)
    ctx.sync()
    for a in accs:
        sums += a[:2 * ctx.n_knodes].cpu().numpy()
    assert np.array_equal(sums.astype(np.uint32), want)


def test_second_stage_inside_the_streaming_kernel(tmp_path, oracle, monkeypatch):
    """sketch_filter_kernel stages the level-0 survivors in LDS and runs the second-stage filter itself, 64 groups at a time
    (no refine_kernel, no group records in global memory: the default; DRPRG_FILTER_FORM=refine is the two-kernel form).  Both
    give the oracle's vector: sparse reads, dense reads whose tiles hold more groups than the stage, reads across slice ends."""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_FILTER_FORM", "fused")
    panel = synth.small_panel(seed=6, n_loci=3, length=900)
    rng = np.random.default_rng(5)
    haps = [synth.sample_haplotype(rng, t).encode() for t in panel.trees]
    dense = _reads_from(rng, haps, 6000, 150)
    sparse = _reads_from(rng, haps + [synth.random_seq(rng, 200000).encode()] * 9, 60000, 150)
    for bases, offs in (dense, sparse):
        ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 2)
        assert cnt["clusters_kept"] > 3000
    if EXPERIMENTAL:  # (the two-kernel form: make EXPERIMENTAL=1)
        monkeypatch.setenv("DRPRG_FILTER_FORM", "refine")
        ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
        _compare(ctx, oracle, sparse[0], sparse[1], 11, 15, True, 2)


@pytest.mark.parametrize("sched,grid", [("static", None), ("100,20,4,8", "3")])
@pytest.mark.parametrize("stage2", ["lds", "l2"])
def test_second_stage_in_lds_or_in_the_l2(tmp_path, oracle, monkeypatch, stage2, sched, grid):
    """Round 6.  The small tier's second stage has two homes: the six bits per code that share the level-0 array in LDS (the default for
    ASCII batches) and a split-block filter of the codes in the L2, probed once per surviving group, with level 0 alone in LDS (the default
    for packed batches).  DRPRG_FILTER_STAGE2 forces one for both formats (_compare maps ASCII and packed): each gives the oracle's vector on
    the ragged / N / lower-case / empty-read case, on dense reads whose tiles overflow the stage, on sparse reads and on 4 kb reads -- with
    one chunk per wave and under the chunk schedule."""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_FILTER_STAGE2", stage2)
    monkeypatch.setenv("DRPRG_FT_SCHED", sched)
    if grid:
        monkeypatch.setenv("DRPRG_FT_GRID", grid)
    panel = synth.small_panel(seed=2)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
    sc = ctx.filter_selfcheck()
    assert sc["codes"] > 0 and sc["shared_array_false_negatives"] == 0
    bases, offs = _ragged_reads(panel)
    _compare(ctx, oracle, bases, offs, 11, 15, True, 2)
    panel = synth.small_panel(seed=6, n_loci=3, length=900)
    rng = np.random.default_rng(15)
    haps = [synth.sample_haplotype(rng, t).encode() for t in panel.trees]
    dense = _reads_from(rng, haps, 40000, 150)
    sparse = _reads_from(rng, haps + [synth.random_seq(rng, 200000).encode()] * 9, 70000, 149)
    for bases, offs in (dense, sparse):
        ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 2)
        assert cnt["clusters_kept"] > 3000
    panel = synth.small_panel(seed=11, n_loci=6, length=1500)
    ctx = _ctx(tmp_path, panel, 11, 15, False, kernel=2)
    gen = synth.HaplotypeGenomes(panel, genome_size=60000, n_hap=4, seed=3)
    bases, offs = synth.sample_long_reads(gen, 3000, seed=4)
    assert _compare(ctx, oracle, bases, offs, 11, 15, False, 2)["clusters_kept"] > 0


# ---- middle tier of the filter (round 3): level 0 on canonical 12-mers in LDS, exact 12-mer bitmap + code filter in the L2 ----------
_SCALED = {}


def _scaled(scale):
    from drprg_amd import synth
    if scale not in _SCALED:
        panel = synth.mtb_scaled_panel(scale, GOLDEN_INDEX_DIR)
        _SCALED[scale] = (panel, synth.HaplotypeGenomes(panel, n_hap=8))
    return _SCALED[scale]


@pytest.mark.parametrize("scale", [2, 4, 8, 16])
def test_scaled_mtb_indexes_take_the_middle_tier(tmp_path, oracle, monkeypatch, scale):
    """The 8d index grown 2-16 fold (30 k - 244 k k-mer nodes): too many k-mers for the all-LDS filter, so `auto` is the filtered
    sequence in its middle-tier form (up to 180 k index records; beyond that the direct sequence is faster and `auto` takes it:
    the 16-fold index is pushed through the middle tier here all the same); coverage and counters equal the oracle's, and the
    direct kernel's candidate form agrees."""
    from drprg_amd import synth
    panel, genomes = _scaled(scale)
    if scale == 16:
        ctx = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE, kernel=0)
        assert ctx.table_tier()["kernel"] == 3 and ctx.table_tier()["l2_filter_bytes"] == 0
        ctx.close()
        monkeypatch.setenv("DRPRG_MID_MAX_RECORDS", "1000000")
    n = 600_000 if scale <= 4 else 300_000
    bases, offs = synth.sample_short_reads(genomes, n, seed=20 + scale)
    ctx = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE, kernel=0)
    tier = ctx.table_tier()
    assert tier["kernel"] == 2 and tier["l2_filter_bytes"] >= 2 << 20 and tier["lds_filter_bytes"] == 128 << 10
    sc = ctx.filter_selfcheck()
    assert sc["codes"] == 2 * ctx.n_records and sc["level0_false_negatives"] == sc["level12_false_negatives"] == sc["stage2_false_negatives"] == 0
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 2, threads=ORACLE_THREADS)
    assert cnt["clusters_kept"] > 2000 * scale
    if not FORCED_GENERIC:
        assert ctx.counters()["leftover_reads"] == 0
    if scale in (2, 16):
        ctx3 = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE, kernel=3)
        _compare(ctx3, oracle, bases, offs, 11, 15, True, 3, threads=ORACLE_THREADS)


def test_scaled_mtb_index_long_reads_middle_tier(tmp_path, oracle):
    from drprg_amd import synth
    panel, genomes = _scaled(4)
    bases, offs = synth.sample_long_reads(genomes, 12_000, seed=31)
    ctx = _ctx(tmp_path, panel, 11, 15, False, genome_size=synth.MTB_GENOME_SIZE, kernel=0)
    assert ctx.table_tier()["l2_filter_bytes"] > 0
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, False, 2, threads=ORACLE_THREADS)
    assert cnt["clusters_kept"] > 300


@pytest.mark.parametrize("w", [11, 14, 1])
def test_middle_tier_on_small_panels(tmp_path, oracle, monkeypatch, w):
    """DRPRG_FORCE_MID_TIER=1 builds the middle-tier tables for an index the all-LDS filter would serve, so the edge cases of the
    small panels run through it: ragged / empty / N-containing / lower-case reads, tiles dense with index k-mers (more groups than
    the stage holds), long noisy reads."""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_FORCE_MID_TIER", "1")
    panel = synth.small_panel(seed=2)
    ctx = _ctx(tmp_path, panel, w, 15, True, kernel=2)
    assert ctx.table_tier()["l2_filter_bytes"] > 0
    rng = np.random.default_rng(0)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=2, seed=3)
    g = gen.haps[0]
    reads = []
    for i in range(12000):
        L = int(rng.choice([0, 1, 14, 15, 24, 25, 26, 40, 150, 151, 300, 4064, 4096, 5000, 8160, 8192]))
        s = int(rng.integers(0, len(g) - L))
        r = g[s:s + L].copy()
        if L and rng.random() < 0.3:
            r[rng.integers(0, L, size=max(1, L // 50))] = ord("N")
        if L and rng.random() < 0.2:
            r = np.frombuffer(r.tobytes().lower(), dtype=np.uint8)
        reads.append(r)
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    _compare(ctx, oracle, np.concatenate(reads), offs, w, 15, True, 2)
    # amplicon-like: every read inside the panel
    dense_panel = synth.small_panel(seed=6, n_loci=3, length=900)
    haps = [synth.sample_haplotype(rng, t).encode() for t in dense_panel.trees]
    bases, offs = _reads_from(rng, haps, 6000, 150)
    ctx = _ctx(tmp_path, dense_panel, w, 15, True, kernel=2)
    cnt = _compare(ctx, oracle, bases, offs, w, 15, True, 2)
    assert cnt["clusters_kept"] > 3000
    # long noisy reads
    lp = synth.small_panel(seed=11, n_loci=6, length=1500)
    ctx = _ctx(tmp_path, lp, w, 15, False, kernel=2)
    gen = synth.HaplotypeGenomes(lp, genome_size=60000, n_hap=4, seed=3)
    bases, offs = synth.sample_long_reads(gen, 1500, seed=3)
    _compare(ctx, oracle, bases, offs, w, 15, False, 2)


def test_native_rccl_communicator_single_rank(tmp_path, oracle):
    """The C-ABI collective of the one-process-per-GPU layout (drprg_hip_comm_unique_id / comm_init_rank / allreduce): a
    communicator of ONE rank on the GPU box (more ranks need more GPUs) -- the id, the init, the grouped in-place
    ncclAllReduce(sum, u32) of both vectors on the context's stream and the teardown all run; the sum over one rank is the
    rank's own vector, which must still equal the oracle's."""
    from drprg_amd import synth
    from drprg_amd.distributed import NativeComm
    panel = synth.small_panel(seed=42)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=0)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 20000, seed=7)
    idx = _oracle_index(oracle, ctx.prg_strings, 11, 15)
    ocov, oprg, _ = _oracle_map(oracle, idx, bases, offs, 11, 15, True)
    ctx.map_host(bases, offs)
    comm = NativeComm(0, 1, 0, exchange=lambda ident: ident)
    comm.allreduce(ctx)
    ctx.sync()
    got, got_prg = ctx.coverage()
    assert np.array_equal(got, ocov) and np.array_equal(got_prg, oprg) and got.sum() > 0
    comm.allreduce(ctx)  # twice: still the same vector with one rank
    assert np.array_equal(ctx.coverage()[0], ocov)
    # north_star: "a single RCCL reduce" -- the context's accumulators are one vector [coverage | reads per PRG]
    import ctypes as C
    from drprg_amd._lib import lib
    buf = C.create_string_buffer(256)
    lib.drprg_hip_reduce_info(ctx._h, buf, len(buf))
    assert buf.value.decode().startswith("rccl: one ncclAllReduce(sum, u32) of %d words" % (got.size + got_prg.size)), buf.value
    # a caller's own packed buffer (n_covg + n_prgs words, what bench.py --comm native passes) takes the same single call;
    # two separate buffers still work (one group of two)
    import torch
    packed = torch.zeros(got.size + got_prg.size, dtype=torch.int32, device="cuda")
    packed[:got.size] = torch.from_numpy(ocov.view(np.int32)).cuda()
    comm.allreduce(ctx, packed.data_ptr(), packed.data_ptr() + 4 * got.size, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.drprg_hip_reduce_info(ctx._h, buf, len(buf))
    assert buf.value.decode().startswith("rccl: one ncclAllReduce")
    assert np.array_equal(packed[:got.size].cpu().numpy().view(np.uint32), ocov)
    a, b = torch.ones(got.size, dtype=torch.int32, device="cuda"), torch.ones(got_prg.size, dtype=torch.int32, device="cuda")
    comm.allreduce(ctx, a.data_ptr(), b.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.drprg_hip_reduce_info(ctx._h, buf, len(buf))
    assert buf.value.decode().startswith("rccl: two grouped") and int(a.sum()) == got.size and int(b.sum()) == got_prg.size
    comm.close()


def test_reopened_direct_context_with_leftover_reads(tmp_path, oracle):
    """A kernel = 3 context whose batches leave reads to the generic pipeline (read_cluster_kernel reads the candidates from the
    tile slices and marks the handled ones in a dense array): open, map, close, open again -- device memory is recycled, the marks
    of the first context must not be taken for marks of the second (the array is cleared when it is allocated and every batch of
    the process uses a mark of its own)."""
    from drprg_amd import synth
    panel = synth.small_panel(seed=5, n_loci=2, length=16000, site_every=50)
    gen = synth.HaplotypeGenomes(panel, genome_size=60000, n_hap=2, seed=3)
    rng = np.random.default_rng(3)
    # 9 kb reads inside a 16 kb locus: more staged hits than the per-read kernel holds -> left over
    reads = []
    for i in range(300):
        h = gen.haps[i % 2]
        s = int(rng.integers(0, len(h) - 9000))
        reads.append(h[s:s + 9000])
    short = synth.sample_short_reads(gen, 4000, seed=11)
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.concatenate(reads + [short[0]])
    offs = np.concatenate([offs, offs[-1] + short[1][1:]])
    for attempt in range(3):
        ctx = _ctx(tmp_path, panel, 11, 15, False, genome_size=60000, kernel=3)
        _compare(ctx, oracle, bases, offs, 11, 15, False, 3)
        if not FORCED_GENERIC:
            assert ctx.counters()["leftover_reads"] > 0
        ctx.close()


def test_packed_reads_on_the_device_and_through_the_ingest(tmp_path, oracle):
    """2-bit packed reads end to end: drprg_hip_pack_device == the host packer; a device-resident packed batch through map_device_packed
    (synchronous and deferred) == the oracle; drprg_hip_map_fastx with drprg_hip_set_input_format(ctx, 1) (the parser threads pack) ==
    the ASCII ingest, on a FASTQ with N, lower case and ragged reads; the reads kept in HBM in the packed form serve discover and
    drprg_hip_map_resident."""
    import torch
    from drprg_amd import Context, synth
    from drprg_amd.pandora import pack_reads
    panel = synth.small_panel(seed=42)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=0)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 40000, seed=7)
    bases = bases.copy()
    rng = np.random.default_rng(1)
    bases[rng.integers(0, bases.size, 300)] = ord("N")
    low = rng.integers(0, bases.size - 30, 200)
    for a in low:
        bases[a:a + 20] |= 0x20
    idx = _oracle_index(oracle, ctx.prg_strings, 11, 15)
    ocov, oprg, _ = _oracle_map(oracle, idx, bases, offs, 11, 15, True)
    words, npos = pack_reads(bases)
    assert npos.size >= 250
    d_bases = torch.from_numpy(bases).cuda()
    d_words = torch.zeros(words.size, dtype=torch.int32, device="cuda")
    d_npos = torch.zeros(1024, dtype=torch.int64, device="cuda")
    n = ctx.pack_device(d_bases.data_ptr(), bases.size, d_words.data_ptr(), d_npos.data_ptr(), 1024)
    assert n == npos.size and np.array_equal(d_words.cpu().numpy().view(np.uint32), words)
    assert np.array_equal(d_npos[:n].cpu().numpy().astype(np.uint64), npos)
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    for deferred in (False, True):
        ctx.reset()
        ctx.map_device_packed(d_words.data_ptr(), d_offs.data_ptr(), len(offs) - 1, int(bases.size), d_npos.data_ptr(), n, deferred=deferred)
        ctx.sync()
        got, got_prg = ctx.coverage()
        assert np.array_equal(got, ocov) and np.array_equal(got_prg, oprg)
    # an ASCII batch at the address a packed batch had is an ASCII batch
    d_alias = d_words.view(torch.uint8)
    n_alias = min(int(d_alias.numel()), int(bases.size)) // 16 * 16
    d_alias[:n_alias] = d_bases[:n_alias]
    k = int(np.searchsorted(offs, n_alias, side="right")) - 1
    ctx.reset()
    ctx.map_device(d_alias.data_ptr(), d_offs.data_ptr(), k, int(offs[k]))
    want_k = _oracle_map(oracle, idx, bases[:int(offs[k])], offs[:k + 1], 11, 15, True)[0]
    assert np.array_equal(ctx.coverage()[0], want_k)
    # ---- the ingest ----
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, bases, offs)
    for threads in (1, 8):
        ctx.reset()
        ctx.set_input_format(True)
        ctx.set_threads(threads)
        ctx.map_fastx(fq)
        got, got_prg = ctx.coverage()
        assert np.array_equal(got, ocov) and np.array_equal(got_prg, oprg), threads
        assert ctx.counters()["reads"] == len(offs) - 1 and ctx.counters()["bases"] == bases.size
    # every kernel sequence takes the packed ingest
    for kernel in (1, 3):
        c2 = _ctx(tmp_path, panel, 11, 15, True, kernel=kernel)
        c2.set_input_format(True)
        c2.set_threads(4)
        c2.map_fastx(fq)
        assert np.array_equal(c2.coverage()[0], ocov), kernel
        c2.close()
    # ---- resident packed reads: discover's read selection and the second mapping pass ----
    ctx.reset()
    ctx.keep_reads(1 << 30)
    ctx.map_fastx(fq)
    info = ctx.resident_info()
    assert info["complete"] and 0 < info["bytes"] < bases.size  # (a quarter of the bases + offsets)
    other = _ctx(tmp_path, panel, 11, 15, True, kernel=0)
    other.map_resident(ctx)
    assert np.array_equal(other.coverage()[0], ocov)


# ---- sketch_filter_kernel's schedule (round 6): chunks handed out inside the kernel, tile shares that follow the batches ---------------
def _ragged_reads(panel, seed=0, n=12000):
    """the mix of test_ragged_and_degenerate_inputs: empty reads, reads shorter than k, N runs, lower case, lengths up to 8192"""
    from drprg_amd import synth
    rng = np.random.default_rng(seed)
    g = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=2, seed=3).haps[0]
    reads = []
    for i in range(n):
        L = int(rng.choice([0, 1, 14, 15, 24, 25, 26, 40, 150, 151, 300, 4064, 4096, 5000, 8160, 8192]))
        s = int(rng.integers(0, len(g) - L))
        r = g[s:s + L].copy()
        if L and rng.random() < 0.3:
            r[rng.integers(0, L, size=max(1, L // 50))] = ord("N")
        if L and rng.random() < 0.2:
            r = np.frombuffer(r.tobytes().lower(), dtype=np.uint8)
        reads.append(r)
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return np.concatenate(reads), offs


@pytest.mark.parametrize("sched", ["static", "180,32,4,8"])
def test_filter_schedule_follows_the_batches(tmp_path, oracle, monkeypatch, sched):
    """VERDICT r05 #2a.  The configs[1] batch (180 M bases: large enough for the host to time the wave classes) mapped eight times on one
    context: with the static schedule the tile shares move from batch to batch (mapper.cpp tune_filter_shares), with the dynamic one the
    waves draw their chunks in whatever order they get to them -- every one of the eight passes must add exactly the oracle's vector.
    Then reset (coverage zero, built-in shares again) and once more."""
    import torch
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_FT_SCHED", sched)
    panel, genomes = _baseline_panel("mtb_8d")
    ctx = _ctx(tmp_path, panel, 11, 15, True, genome_size=synth.MTB_GENOME_SIZE)
    bases, offs = _baseline_reads("mtb_8d", False, 1_200_000)
    idx = _oracle_index(oracle, ctx.prg_strings, 11, 15)
    ocov, oprg, ocnt = _oracle_map(oracle, idx, bases, offs, 11, 15, True, threads=ORACLE_THREADS)
    dev = torch.device("cuda", 0)
    tb, to = torch.from_numpy(np.ascontiguousarray(bases)).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    ctx.reset()
    seen = []
    for i in range(8):
        ctx.map_device_async(tb.data_ptr(), to.data_ptr(), len(offs) - 1, int(offs[-1]))
        if i >= 1:
            seen.append(tuple(ctx.filter_schedule()["round0_class_shares_per_256"]))  # (synchronises: of the batch just completed)
    cov, prg = ctx.coverage()
    assert np.array_equal(cov.astype(np.uint64), 8 * ocov.astype(np.uint64)) and np.array_equal(prg.astype(np.uint64), 8 * oprg.astype(np.uint64))
    info = ctx.filter_schedule()
    assert ctx.counters()["kernel"] == 2 and ctx.counters()["hits"] == 8 * ocnt["hits"]
    if sched == "static":
        assert info["form"] == "static" and (not ONE_LANE or len(set(seen)) > 1), seen  # the shares did move
    elif ONE_LANE:
        assert info["form"] == "dynamic" and info["slices_per_workgroup"] > 16 and len(set(seen)) == 1, (info, seen)
    ctx.reset()
    ctx.map_device(tb.data_ptr(), to.data_ptr(), len(offs) - 1, int(offs[-1]))
    cov, prg = ctx.coverage()
    assert np.array_equal(cov, ocov) and np.array_equal(prg, oprg)
    # ... and the same through the packed form, three passes
    from drprg_amd.pandora import pack_reads
    words, npos = pack_reads(bases)
    ctx.reset()
    for _ in range(3):
        ctx.map_host_packed(words, offs, npos)
    cov, prg = ctx.coverage()
    assert np.array_equal(cov.astype(np.uint64), 3 * ocov.astype(np.uint64)) and np.array_equal(prg.astype(np.uint64), 3 * oprg.astype(np.uint64))


ONE_LANE = int(os.environ.get("DRPRG_HIP_LANES", "1") or 1) == 1  # (a batch cut into read ranges on concurrent streams keeps one chunk per
# wave: the host has no tile numbers for a range -- the suite still runs that way, tools/flaky_record.sh, without the assertions on the form)


def _schedule_forms(ctx, bases, offs):
    """the schedule sketch_filter_kernel ran this batch with, from ASCII and from the packed words: ("static" | "dynamic") x 2"""
    from drprg_amd.pandora import pack_reads
    ctx.reset()
    ctx.map_host(bases, offs)
    a = ctx.filter_schedule()
    words, npos = pack_reads(bases)
    ctx.reset()
    ctx.map_host_packed(words, offs, npos)
    p = ctx.filter_schedule()
    assert a["form"] == "static" or a["slices_per_workgroup"] > 16
    return a["form"], p["form"]


@pytest.mark.parametrize("share", ["2.2,0.35,0.35,2.2", "0.35,2.2,2.2,0.35"])
@pytest.mark.parametrize("sched,grid", [("static", None), ("128,32,4,8", "4"), ("static", "3")])
def test_extreme_tile_shares(tmp_path, oracle, monkeypatch, share, sched, grid):
    """VERDICT r05 #2b.  The wave classes' shares at both clamps of the host's tuning (DRPRG_FT_SHARE), on the ragged / N / lower-case /
    empty-read case and on 4 kb reads, ASCII and packed (_compare maps both): any shares give the oracle's vector -- with one chunk per
    wave, and with the chunk schedule on top of them (few workgroups, so that these small batches have tiles enough for one)."""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_FT_SHARE", share)
    monkeypatch.setenv("DRPRG_FT_SCHED", sched)
    if grid:
        monkeypatch.setenv("DRPRG_FT_GRID", grid)
    want = ("dynamic", "dynamic") if sched != "static" else ("static", "static")
    panel = synth.small_panel(seed=2)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
    bases, offs = _ragged_reads(panel)
    _compare(ctx, oracle, bases, offs, 11, 15, True, 2)
    assert not ONE_LANE or _schedule_forms(ctx, bases, offs) == want
    panel = synth.small_panel(seed=11, n_loci=6, length=1500)
    ctx = _ctx(tmp_path, panel, 11, 15, False, kernel=2)
    gen = synth.HaplotypeGenomes(panel, genome_size=60000, n_hap=4, seed=3)
    bases, offs = synth.sample_long_reads(gen, 3000, seed=3)
    cnt = _compare(ctx, oracle, bases, offs, 11, 15, False, 2)
    assert cnt["clusters_kept"] > 0 and (not ONE_LANE or _schedule_forms(ctx, bases, offs) == want)


@pytest.mark.parametrize("grid,sched", [("1", "100,20,4,8"), ("3", "100,20,4,8"), ("7", "200,17,4,8"), ("5", "60,64,5,8"), ("2", "250,1024,4,8")])
def test_chunk_schedule_edges(tmp_path, oracle, monkeypatch, grid, sched):
    """VERDICT r05 #2c.  The chunk schedule with chunks small enough that a workgroup's last chunk is a partial one, the workgroups' ranges
    differ by a tile, the batch ends in tiles that do not lie wholly inside the buffer, and most waves never get to draw a ticket (one
    workgroup; seven): dense reads (slices that fill up), sparse reads, the middle tier, every case in both input formats."""
    from drprg_amd import synth
    monkeypatch.setenv("DRPRG_FT_GRID", grid)
    monkeypatch.setenv("DRPRG_FT_SCHED", sched)
    panel = synth.small_panel(seed=6, n_loci=3, length=900)
    rng = np.random.default_rng(5)
    haps = [synth.sample_haplotype(rng, t).encode() for t in panel.trees]
    dense = _reads_from(rng, haps, 60000 + 37 * int(grid), 150)
    sparse = _reads_from(rng, haps + [synth.random_seq(rng, 200000).encode()] * 9, 90000 + 11 * int(grid), 151)
    forms = []
    for bases, offs in (dense, sparse):
        ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
        cnt = _compare(ctx, oracle, bases, offs, 11, 15, True, 2)
        assert cnt["clusters_kept"] > 3000
        forms.append(_schedule_forms(ctx, bases, offs))
    assert not ONE_LANE or all(f[0] == "dynamic" for f in forms), forms
    if ONE_LANE and sched != "60,64,5,8":  # (a quarter of the tiles in round 0: the packed form's 64 positions per lane leave too few tiles per wave for it)
        assert all(f[1] == "dynamic" for f in forms), forms
    # the middle tier (its own instantiation of the kernel) and w = 14 (another window) through the same schedule
    monkeypatch.setenv("DRPRG_FORCE_MID_TIER", "1")
    ctx = _ctx(tmp_path, panel, 14, 15, True, kernel=2)
    _compare(ctx, oracle, sparse[0], sparse[1], 14, 15, True, 2)
    assert not ONE_LANE or _schedule_forms(ctx, sparse[0], sparse[1])[0] == "dynamic"
    monkeypatch.delenv("DRPRG_FORCE_MID_TIER")
    # k = 13 (the forms without level 0: two workgroups per CU, another LDS layout for the counter)
    ctx = _ctx(tmp_path, panel, 16, 13, True, kernel=2)
    _compare(ctx, oracle, sparse[0], sparse[1], 16, 13, True, 2)
    assert not ONE_LANE or _schedule_forms(ctx, sparse[0], sparse[1])[0] == "dynamic"


def test_offsets_must_span_the_batch_under_a_chunk_schedule(tmp_path, oracle, monkeypatch):
    """The chunk schedule is made by the host for the tiles of the WHOLE batch (it cannot read the offsets, they live on the device): a
    device batch whose offsets do not span [0, n_bases) is refused -- loudly, nothing mapped -- and the context maps the next batch as if
    nothing had happened.  (One chunk per wave, the schedule of small batches, takes its window from the offsets and does not care.)"""
    import torch
    from drprg_amd import DependencyError, synth
    if not ONE_LANE:
        pytest.skip("read ranges on concurrent streams keep one chunk per wave")
    monkeypatch.setenv("DRPRG_FT_GRID", "2")
    monkeypatch.setenv("DRPRG_FT_SCHED", "128,32,4,8")
    panel = synth.small_panel(seed=4)
    ctx = _ctx(tmp_path, panel, 11, 15, True, kernel=2)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 20000, seed=8)
    dev = torch.device("cuda", 0)
    tb = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
    bad = offs.astype(np.int64).copy()
    bad[0] = bad[40]  # (the first 40 reads empty: still ascending, but the window starts 6000 bases in)
    bad[:40] = bad[40]
    tbad, tgood = torch.from_numpy(bad).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    ctx.reset()
    with pytest.raises(DependencyError, match="offsets"):
        ctx.map_device(tb.data_ptr(), tbad.data_ptr(), len(offs) - 1, int(offs[-1]))
    ctx.reset()
    ctx.map_device(tb.data_ptr(), tgood.data_ptr(), len(offs) - 1, int(offs[-1]))
    cov, prg = ctx.coverage()
    assert ctx.filter_schedule()["form"] == ("dynamic" if ONE_LANE else "static")
    idx = _oracle_index(oracle, ctx.prg_strings, 11, 15)
    ocov, oprg, _ = _oracle_map(oracle, idx, bases, offs, 11, 15, True)
    assert np.array_equal(cov, ocov) and np.array_equal(prg, oprg)
