"""Coverage model, best path and presence rule (drprg_amd/csrc/params.cpp) against the oracle's separate statement
(oracle/oracle_params.c, oracle_index.c): everything `pandora map` computes between its read loop and the VCF.

PARITY UNPINNED on the pandora side: no reference fixture exercises these functions (pandora is an external binary,
/root/reference/src/lib.rs:580-642).  The one pin the reference holds is recorded at the bottom: the e of each of the seven
fixture VCFs is a positive integer, i.e. a value the estimator can produce."""
import ctypes as C
import os

import numpy as np
import pytest

from util import GOLDEN, Oracle, _p


@pytest.fixture(scope="module")
def orc():
    o = Oracle()
    L = o.lib
    L.orc_estimate_parameters.restype = None
    L.orc_estimate_parameters.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_int, C.c_void_p]
    L.orc_kmer_log_prob.restype = C.c_float
    L.orc_kmer_log_prob.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_prob_threshold.restype = C.c_int
    L.orc_prob_threshold.argtypes = [C.c_void_p, C.c_int64]
    L.orc_max_path.restype = C.c_int64
    L.orc_max_path.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int64]
    L.orc_mode.restype = C.c_uint32
    L.orc_mode.argtypes = [C.c_void_p, C.c_int64]
    L.orc_path_coverage_too_low.restype = C.c_int
    L.orc_path_coverage_too_low.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
    L.orc_kg_base_coverage.restype = C.c_int64
    L.orc_kg_base_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    return o


def _both_estimates(orc, kc, clusters, loci, global_covg, k, e_rate, bin_):
    from drprg_amd._lib import lib
    kc = np.ascontiguousarray(kc, np.uint32)
    a = np.zeros(10)
    orc.lib.orc_estimate_parameters(_p(kc), len(kc), clusters, loci, global_covg, k, e_rate, bin_, _p(a))
    b = (C.c_double * 10)()
    assert lib.drprg_hip_estimate_parameters(_p(kc), len(kc), clusters, loci, global_covg, k, e_rate, bin_, b) == 0
    return a, np.array(list(b))


def _histogram_cases():
    rng = np.random.default_rng(11)
    cases = []
    # (name, k-mer coverages, clusters, loci, global coverage, --bin, expected branch)
    nb = rng.negative_binomial(8, 8 / (8 + 60.0), size=20000)                                       # over-dispersed around 60
    cases.append(("negative binomial", nb, 18 * 400, 18, 60, 0, 2))
    cases.append(("negative binomial with an error peak", np.concatenate([nb, rng.poisson(0.3, 3000)]), 18 * 400, 18, 60, 0, 2))
    po = rng.poisson(40, size=20000)                                                                 # variance ~ mean: binomial chosen
    cases.append(("poisson-like: switches to binomial", np.concatenate([po, rng.poisson(0.2, 4000)]), 18 * 300, 18, 45, 0, 1))
    cases.append(("asked for binomial", np.concatenate([nb, rng.poisson(0.3, 3000)]), 18 * 400, 18, 60, 1, 1))
    cases.append(("binomial asked, too shallow", rng.poisson(12, 5000), 18 * 40, 18, 14, 1, 3))
    cases.append(("few reads per locus", nb, 18 * 20, 18, 60, 0, 3))
    cases.append(("almost nothing mapped", rng.poisson(0.5, 2000), 40, 10, 1, 0, 3))
    cases.append(("under-dispersed above covg / 10, refit from 2", np.concatenate([np.full(5000, 50), rng.integers(0, 3, 800)]), 18 * 100, 18, 55, 0, None))
    cases.append(("saturated counters are ignored", np.concatenate([nb, np.full(100, 65535 * 2)]), 18 * 400, 18, 60, 0, 2))
    cases.append(("global coverage 2600x wraps the 8-bit threshold", rng.negative_binomial(8, 8 / (8 + 700.0), size=5000), 18 * 4000, 18, 2600, 0, None))
    cases.append(("no locus at all", np.zeros(0, np.int64), 0, 0, 33, 0, 0))
    return cases


@pytest.mark.parametrize("case", _histogram_cases(), ids=lambda c: c[0])
def test_estimate_parameters_equals_the_oracle(orc, case):
    name, kc, clusters, loci, gcov, bin_, branch = case
    a, b = _both_estimates(orc, kc, clusters, loci, gcov, 15, 0.11, bin_)
    assert np.allclose(a, b, rtol=1e-12, atol=0), (name, a, b)
    assert a[0] >= 1 and a[0] == int(a[0])
    if branch is not None:
        assert int(a[5]) == branch, (name, a)
    if int(a[5]) == 2:  # e = floor(mean of the histogram from covg / 10 upwards), negative binomial p = mean / var in (0, 1)
        assert a[0] == int(a[6]) and 0 < a[3] < 1 and a[4] > 0
    if int(a[5]) == 1:
        assert a[1] == 1


def test_kmer_log_probabilities_and_threshold_equal_the_oracle(orc):
    from drprg_amd._lib import lib
    rng = np.random.default_rng(5)
    out = C.c_float()
    lp = []
    for use_bin in (0, 1):
        for _ in range(400):
            fwd, rev = int(rng.integers(0, 200)), int(rng.integers(0, 200))
            reads = int(rng.integers(1, 500))
            nb_p, nb_r, bin_p = float(rng.uniform(0.01, 0.9)), float(rng.uniform(0.5, 30)), float(rng.uniform(0.05, 0.99))
            want = orc.lib.orc_kmer_log_prob(use_bin, nb_p, nb_r, bin_p, fwd, rev, reads)
            assert lib.drprg_hip_kmer_log_prob(use_bin, nb_p, nb_r, bin_p, fwd, rev, reads, C.byref(out)) == 0
            assert np.isclose(out.value, want, rtol=2e-6, atol=1e-6), (use_bin, fwd, rev, reads, out.value, want)
            lp.append(want)
    # thresholds: two peaks, one peak, nothing in range
    for arr in (np.concatenate([rng.normal(-3, 1, 5000), rng.normal(-60, 4, 800)]), rng.normal(-4, 1.5, 3000), np.full(10, -500.0),
                np.array(lp)):
        arr = np.ascontiguousarray(arr, np.float32)
        t = C.c_int()
        assert lib.drprg_hip_prob_threshold(_p(arr), len(arr), C.byref(t)) == 0
        assert t.value == orc.lib.orc_prob_threshold(_p(arr), len(arr))
        assert -200 <= t.value <= 0


def _panel_ctx(tmp_path, seed):
    from drprg_amd import Context, synth
    panel = synth.small_panel(seed=seed, n_loci=3, length=600, site_every=30)
    prg = str(tmp_path / f"dr{seed}.prg")
    panel.write(prg)
    return panel, Context(prg, 11, 15, device=-1, from_files=False)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_max_path_base_coverage_and_presence_rule_equal_the_oracle(tmp_path, orc, seed):
    """Random coverages on the k-mer graphs of nested / indel panels: the product's best path, the per-base coverage along it and the
    drop decision must be the oracle's (its own k-mer graph, its own dynamic programme).  Includes all-zero coverage (every mean
    ties: the longer path wins), a coverage that follows one haplotype, and a short averaging window."""
    from drprg_amd._lib import lib
    panel, ctx = _panel_ctx(tmp_path, seed)
    rng = np.random.default_rng(seed)
    for prg_i, prg_string in enumerate(panel.prgs):
        g = orc.lib.orc_index_prg(prg_string.encode(), 11, 15)
        try:
            sk = orc.sketch_prg(prg_string, 11, 15)
            n = sk["n_nodes"]
            ef = np.ascontiguousarray(sk["edges"][:, 0], np.uint32)
            et = np.ascontiguousarray(sk["edges"][:, 1], np.uint32)
            for mode in ("zero", "random", "sparse", "deep"):
                if mode == "zero":
                    fwd = rev = np.zeros(n, np.uint32)
                elif mode == "random":
                    fwd, rev = rng.poisson(20, n).astype(np.uint32), rng.poisson(20, n).astype(np.uint32)
                elif mode == "sparse":
                    fwd, rev = (rng.random(n) < 0.1).astype(np.uint32) * 2, np.zeros(n, np.uint32)
                else:
                    fwd, rev = rng.poisson(60, n).astype(np.uint32), rng.poisson(55, n).astype(np.uint32)
                    fwd[rng.random(n) < 0.5] = 0
                fwd[0] = fwd[-1] = rev[0] = rev[-1] = 0
                logp = np.zeros(n, np.float32)
                for i in range(1, n - 1):
                    logp[i] = orc.lib.orc_kmer_log_prob(0, 0.1, 5.0, 0.8, int(fwd[i]), int(rev[i]), 100)
                for max_avg in (100, 7):
                    want = np.zeros(n, np.uint32)
                    m = orc.lib.orc_max_path(n, len(ef), _p(ef), _p(et), _p(logp), -25, max_avg, _p(want), n)
                    got = np.zeros(n, np.uint32)
                    ng = C.c_uint64()
                    assert lib.drprg_hip_max_path(ctx._h, prg_i, _p(logp), -25, max_avg, _p(got), n, C.byref(ng)) == 0
                    assert ng.value == m and np.array_equal(got[:m], want[:m]), (mode, max_avg)
                    assert m > 0 and np.all(np.diff(want[:m].astype(np.int64)) > 0)  # a source-to-sink walk in topological order
                    # per-base coverage along it + the rule
                    total = (fwd + rev).astype(np.uint32)
                    cap = 1 << 16
                    ob = np.zeros(cap, np.uint32)
                    no = orc.lib.orc_kg_base_coverage(g, _p(want), m, _p(total), _p(ob), cap)
                    c2 = np.ascontiguousarray(np.stack([fwd, rev], axis=1).reshape(-1), np.uint32)
                    pb = np.zeros(cap, np.uint32)
                    npb = C.c_uint64()
                    assert lib.drprg_hip_path_base_coverage(ctx._h, prg_i, _p(want), m, _p(c2), _p(pb), cap, C.byref(npb)) == 0
                    assert npb.value == no and np.array_equal(pb[:no], ob[:no]), mode
                    for gcov in (5, 21, 80):
                        assert lib.drprg_hip_path_coverage_too_low(_p(pb), no, gcov) == orc.lib.orc_path_coverage_too_low(_p(ob), no, gcov)
                    if mode in ("zero", "sparse"):
                        assert lib.drprg_hip_path_coverage_too_low(_p(pb), no, 80) == 1 and lib.drprg_hip_path_coverage_too_low(_p(pb), no, 20) == 0
                    if mode == "random":
                        assert lib.drprg_hip_path_coverage_too_low(_p(pb), no, 80) == 0
        finally:
            orc.lib.orc_kg_free(g)


def test_mode_is_pandoras(orc):
    for v, want in (([5, 5, 1, 1, 9], 1), ([1, 2, 3], 0), ([7], 0), ([0, 0, 4, 4, 4], 4), ([3, 3], 3)):
        a = np.array(v, np.uint32)
        assert orc.lib.orc_mode(_p(a), len(a)) == want


def test_genotype_runs_the_model_and_drops_a_bare_locus(tmp_path, orc):
    """drprg_hip_genotype end to end on a host-only context fed a coverage vector: a deep sample (global coverage 60) in which one
    locus has clusters but an almost bare best path -> that locus gets no ##contig line (drprg reports it absent,
    /root/reference/src/predict.rs:757-765); in a shallow sample (global coverage <= 20) the same vector keeps it."""
    from drprg_amd import Context, synth
    panel = synth.small_panel(seed=9, n_loci=3, length=700)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    ctx = Context(prg, 11, 15, device=-1, from_files=False)
    rng = np.random.default_rng(2)
    covg = np.zeros(2 * ctx.n_knodes, np.uint32)
    prg_reads = np.zeros(ctx.n_prgs, np.uint32)
    exp = ctx.export_index()
    base = exp["knode_base"]
    for i in range(3):
        n = int(base[i + 1] - base[i])
        if i < 2:
            covg[2 * base[i]:2 * base[i + 1]] = rng.negative_binomial(6, 6 / (6 + 30.0), 2 * n)
            prg_reads[i] = 400
        else:  # one stray cluster of 13 hits
            idx = 2 * (int(base[i]) + 5 + np.arange(13))
            covg[idx] = 1
            prg_reads[i] = 1
    for gsize, want_present in ((20_000, 2), (200_000, 3)):  # 1.2 Mbases mapped: 60x / 6x
        ctx.set_opts(illumina=True, genome_size=gsize)
        ctx.set_coverage(covg, prg_reads, 1_200_000)
        info = ctx.genotype(genes, str(tmp_path / f"out{gsize}.vcf"))
        assert info["loci_present"] == want_present
        contigs = [l for l in open(tmp_path / f"out{gsize}.vcf") if l.startswith("##contig")]
        assert len(contigs) == want_present
        m = ctx.coverage_model()
        assert m["exp_depth_covg"] == info["exp_depth_covg"] >= 1 and m["dropped_low_coverage"] == 3 - want_present


# ---- the pin the reference holds -----------------------------------------------------------------------------------------------
FIXTURE_E = {"in.vcf": 96, "in2.vcf": 238, "in3.vcf": 241, "in4.vcf": 73, "ERR4796933.pandora.vcf": 72, "SRR6824468.vcf": 248,
             "ERR2510634.drprg.vcf": 17}


def test_fixture_e_values_are_integers_the_estimator_can_produce():
    """the e of each of the seven pandora VCFs under /root/reference/tests/cases/predict (copies in
    tests/golden/downstream) can be read off its records -- a record without coverage has LIKELIHOOD = -2e for every allele --
    and is a positive integer: exactly what every branch of estimate_parameters returns (a histogram position, or a floored mean,
    at least 1).  Nothing more is pinned: the fixtures hold no k-mer coverage histogram, no read counts and no global coverage, so
    which branch produced the value cannot be recovered from them."""
    import re
    found = {}
    d = os.path.join(GOLDEN, "downstream")
    for name, want in FIXTURE_E.items():
        path = os.path.join(d, name)
        if not os.path.exists(path):
            continue
        es = set()
        for line in open(path):
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            keys = f[8].split(":")
            vals = dict(zip(keys, f[9].split(":")))
            if "LIKELIHOOD" not in vals or "MEAN_FWD_COVG" not in vals:
                continue
            cov = [int(x) for x in re.split(",", vals["MEAN_FWD_COVG"])] + [int(x) for x in re.split(",", vals["MEAN_REV_COVG"])]
            gaps = [float(x) for x in vals["GAPS"].split(",")]
            if any(cov) or any(g != 1 for g in gaps):
                continue
            for lik in vals["LIKELIHOOD"].split(","):
                es.add(-float(lik) / 2)
        if es:
            found[name] = es
            assert all(abs(e - round(e)) < 1e-3 and round(e) >= 1 for e in es), (name, es)
            assert want in {round(e) for e in es}, (name, es)
    assert found, "no fixture VCF with a zero-coverage record found under tests/golden/downstream"
