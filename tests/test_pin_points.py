"""The a-7 pin points: pandora's per-read clustering -- define_clusters, filter_clusters, the hit and cluster
orders -- and the locus-presence rule are restated from the 0.9.x line of pandora; the reference pins the binary 0.10.0-alpha.0.1
(/root/reference/justfile:16-23; `-K` in /root/reference/src/predict.rs:243-245 is a flag of that line; locus absence is read off the
VCF header, /root/reference/src/predict.rs:757-765) and holds no read to check them on.  This file names the oracle functions a
future pin -- one pandora run on known reads -- would replace, and holds today's rules as known answers, so that such a pin shows up
as a diff in ONE place: oracle/oracle.c orc_sort_hits / orc_define_clusters / orc_filter_clusters, oracle/oracle_params.c
orc_path_coverage_too_low ("version risk")."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_POINTS = {  # function -> (file, the HIP code that has to follow a change)
    "orc_sort_hits": ("oracle/oracle.c", ["read_cluster.hip", "cluster.hip"]),
    "orc_define_clusters": ("oracle/oracle.c", ["read_cluster.hip", "cluster.hip"]),
    "orc_filter_clusters": ("oracle/oracle.c", ["read_cluster.hip", "cluster.hip"]),
    "orc_path_coverage_too_low": ("oracle/oracle_params.c", ["params.cpp"]),
}


class Hit(C.Structure):
    _fields_ = [("prg", C.c_uint32), ("knode", C.c_uint32), ("pos", C.c_uint32), ("fwd", C.c_uint8)]


class Cluster(C.Structure):
    _fields_ = [("first", C.c_int), ("n", C.c_int), ("prg", C.c_uint32), ("first_pos", C.c_uint32), ("last_pos", C.c_uint32),
                ("fwd", C.c_uint8), ("alive", C.c_uint8)]


def _lib(oracle):
    lib = oracle.lib
    lib.orc_sort_hits.argtypes = [C.POINTER(Hit), C.c_int64]
    lib.orc_sort_hits.restype = None
    lib.orc_define_clusters.argtypes = [C.POINTER(Hit), C.c_int64, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(Cluster), C.c_int64]
    lib.orc_define_clusters.restype = C.c_int64
    lib.orc_filter_clusters.argtypes = [C.POINTER(Cluster), C.c_int64]
    lib.orc_filter_clusters.restype = None
    return lib


def _clusters(oracle, hits, read_len=150, w=11, max_diff=31, fraction=0.4925, min_cluster_size=3, min_path=(6, 6, 6)):
    """hits: (prg, fwd, pos) in any order -> the clusters define_clusters keeps, then filter_clusters' verdict (shortest k-mer path 6:
    floor(6 * 0.4925) = 2, so the size threshold is min_cluster_size unless a test says otherwise)"""
    lib = _lib(oracle)
    arr = (Hit * len(hits))(*[Hit(p, i, pos, f) for i, (p, f, pos) in enumerate(hits)])
    lib.orc_sort_hits(arr, len(hits))
    mp = (C.c_uint32 * len(min_path))(*min_path)
    out = (Cluster * 64)()
    nc = lib.orc_define_clusters(arr, len(hits), read_len, w, max_diff, fraction, min_cluster_size, mp, out, 64)
    lib.orc_filter_clusters(out, nc)
    return [(c.prg, int(c.fwd), c.n, c.first_pos, c.last_pos, int(c.alive)) for c in out[:nc]]


def test_the_pin_points_exist_and_design_names_them():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "0.10.0-alpha.0.1" in design and "version risk" in design.lower()
    for fn, (path, hip) in PIN_POINTS.items():
        src = open(os.path.join(ROOT, path)).read()
        assert re.search(rf"\b{fn}\s*\(", src), fn
        assert fn in design, f"DESIGN.md section 4 must name {fn}"
        for h in hip:
            assert os.path.exists(os.path.join(ROOT, "drprg_amd", "csrc", h))
    assert open(os.path.join(ROOT, "oracle", "oracle.c")).read().count("PIN POINT") >= 4


def test_hit_order(oracle):
    """prg, forward first, position, k-mer node"""
    lib = _lib(oracle)
    hits = [(1, 0, 5), (0, 0, 9), (0, 1, 30), (0, 1, 2), (1, 1, 7)]
    arr = (Hit * len(hits))(*[Hit(p, i, pos, f) for i, (p, f, pos) in enumerate(hits)])
    lib.orc_sort_hits(arr, len(hits))
    assert [(h.prg, h.fwd, h.pos) for h in arr] == [(0, 1, 2), (0, 1, 30), (0, 0, 9), (1, 1, 7), (1, 0, 5)]


def test_define_clusters_cuts_and_thresholds(oracle):
    run = lambda start, n, prg=0, fwd=1, step=6: [(prg, fwd, start + step * i) for i in range(n)]
    # a gap of exactly max_diff does not cut, max_diff + 1 does
    assert [c[2] for c in _clusters(oracle, run(0, 4) + run(18 + 31, 4))] == [8]
    assert [c[2] for c in _clusters(oracle, run(0, 4) + run(18 + 32, 4))] == [4, 4]
    # kept iff MORE hits than the threshold: min_cluster_size 3 -> three hits are dropped, four kept
    assert _clusters(oracle, run(0, 3)) == [] and len(_clusters(oracle, run(0, 4))) == 1
    # the length-based share: floor(min(shortest path, 2 * len / (w + 1)) * fraction); a 150-base read expects 25, a PRG of 12: floor(12 * 0.4925) = 5
    assert _clusters(oracle, run(0, 5), min_path=(12,)) == [] and len(_clusters(oracle, run(0, 6), min_path=(12,))) == 1
    assert len(_clusters(oracle, run(0, 12, step=2), min_path=(100,))) == 0 and len(_clusters(oracle, run(0, 13, step=2), min_path=(100,))) == 1  # floor(25 * 0.4925) = 12
    # another PRG, or the other strand, starts a cluster of its own
    got = _clusters(oracle, run(0, 4) + run(3, 4, prg=1) + run(200, 4, fwd=0))
    assert sorted((c[0], c[1], c[2]) for c in got) == [(0, 0, 4), (0, 1, 4), (1, 1, 4)]


def test_filter_clusters_sweep(oracle):
    run = lambda start, n, prg=0, fwd=1, step=6: [(prg, fwd, start + step * i) for i in range(n)]
    alive = lambda cl: sorted((c[0], c[1], c[2]) for c in cl if c[5])
    # same PRG, other strand: the smaller dies, wherever it lies
    assert alive(_clusters(oracle, run(0, 6) + run(100, 4, fwd=0))) == [(0, 1, 6)]
    # equal sizes: the earlier one in cluster order (first position) stays
    assert alive(_clusters(oracle, run(0, 4) + run(100, 4, fwd=0))) == [(0, 1, 4)]
    # two PRGs over the same stretch of the read: the one that ends at or before the other's end is "contained"; the smaller dies
    assert alive(_clusters(oracle, run(0, 8) + run(3, 5, prg=1))) == [(0, 1, 8)]
    assert alive(_clusters(oracle, run(0, 5) + run(3, 8, prg=1, step=3))) == [(1, 1, 8)]
    # ... but a cluster that starts inside the previous one and ends BEHIND it survives next to it (overlap alone does not kill)
    assert alive(_clusters(oracle, run(0, 6) + run(20, 6, prg=1))) == [(0, 1, 6), (1, 1, 6)]
    # same first position: the larger comes first in cluster order, so it is the "previous" the smaller is measured against
    assert alive(_clusters(oracle, run(0, 5) + run(0, 7, prg=1, step=4))) == [(1, 1, 7)]


def test_presence_rule(oracle):
    """a locus with a cluster is reported absent iff covg > 20 and the best path's per-base coverage has mode < 3 and mean < 3
    (oracle_params.c orc_path_coverage_too_low; the only consumer: /root/reference/src/predict.rs:757-765 via the ##contig lines)"""
    lib = oracle.lib
    lib.orc_path_coverage_too_low.argtypes = [C.POINTER(C.c_uint32), C.c_int64, C.c_uint32]
    lib.orc_path_coverage_too_low.restype = C.c_int

    def low(cov, global_covg):
        a = np.ascontiguousarray(cov, np.uint32)
        return bool(lib.orc_path_coverage_too_low(a.ctypes.data_as(C.POINTER(C.c_uint32)), len(a), global_covg))

    assert low([0] * 50 + [1, 1, 2], 21) and not low([0] * 50 + [1, 1, 2], 20)      # only deep samples (covg > 20)
    assert not low([3] * 30 + [0] * 10, 40)                                         # mode 3: present
    assert low([0] * 30 + [9] * 9, 40) and not low([0] * 30 + [9] * 15, 40)         # mode 0: the mean (2.08 / 3.0) decides
