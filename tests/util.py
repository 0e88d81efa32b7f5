"""Test helpers: ctypes binding of the CPU oracle (oracle/liboracle.so) and small workload builders.

The oracle is the checker only: nothing under drprg_amd/ imports this file or oracle/.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def ensure_built():
    need = [ORACLE_SO, os.path.join(ROOT, "drprg_amd", "lib", "libdrprg_hip.so"), os.path.join(ROOT, "drprg_amd", "bin", "pandora")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-j4"], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        ensure_built()
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.orc_hash64.restype = C.c_uint64
        L.orc_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_sketch.restype = C.c_int64
        L.orc_sketch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_map_reads.restype = C.c_int
        L.orc_map_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint32,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_allele_stats.restype = None
        L.orc_allele_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_likelihood.restype = C.c_double
        L.orc_likelihood.argtypes = [C.c_double] * 5
        L.orc_genotype.restype = None
        L.orc_genotype.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                   C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_exp_depth_covg.restype = C.c_uint32
        L.orc_exp_depth_covg.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]

    def sketch(self, seq, w, k):
        seq = np.frombuffer(seq if isinstance(seq, bytes) else seq.encode(), dtype=np.uint8)
        cap = max(16, len(seq))
        h = np.zeros(cap, np.uint64)
        p = np.zeros(cap, np.uint32)
        s = np.zeros(cap, np.uint8)
        n = self.lib.orc_sketch(_p(seq), len(seq), w, k, _p(h), _p(p), _p(s), cap)
        return h[:n].copy(), p[:n].copy(), s[:n].copy()

    def map_reads(self, bases, offsets, idx, w, k, max_diff, fraction, min_cluster_size):
        """idx = Context.export_index().  Returns (covg, prg_reads, counters)."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n_knodes = int(idx["knode_base"][-1])
        covg = np.zeros(2 * n_knodes, np.uint32)
        prg_reads = np.zeros(len(idx["min_path_len"]), np.uint32)
        counters = np.zeros(8, np.uint64)
        if bases.size == 0:
            bases = np.zeros(1, np.uint8)
        rc = self.lib.orc_map_reads(_p(bases), _p(offsets), len(offsets) - 1, w, k, max_diff, fraction, min_cluster_size,
                                    _p(idx["keys"]), len(idx["keys"]), _p(idx["rec_off"]), _p(idx["rec_prg"]),
                                    _p(idx["rec_knode"]), _p(idx["rec_strand"]), _p(idx["min_path_len"]), _p(covg),
                                    _p(prg_reads), _p(counters))
        assert rc == 0
        return covg, prg_reads, dict(zip(("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept"),
                                         (int(x) for x in counters)))

    def allele_stats(self, fwd, rev, min_kmer_covg):
        fwd = np.ascontiguousarray(fwd, np.uint32)
        rev = np.ascontiguousarray(rev, np.uint32)
        out = np.zeros(6, np.uint32)
        gaps = C.c_double()
        self.lib.orc_allele_stats(_p(fwd), _p(rev), len(fwd), min_kmer_covg, _p(out), C.byref(gaps))
        return [int(x) for x in out], gaps.value

    def genotype(self, mean_fwd, mean_rev, gaps, e, eps=0.01):
        mf = np.ascontiguousarray(mean_fwd, np.uint32)
        mr = np.ascontiguousarray(mean_rev, np.uint32)
        g = np.ascontiguousarray(gaps, np.float64)
        lik = np.zeros(len(mf), np.float64)
        gt = C.c_int()
        conf = C.c_double()
        self.lib.orc_genotype(_p(mf), _p(mr), _p(g), len(mf), e, eps, _p(lik), C.byref(gt), C.byref(conf))
        return lik, gt.value, conf.value


def cluster_fraction(error_rate, k):
    return 0.5 / np.exp(error_rate * k)


def map_params(k, illumina):
    """(max_diff, error_rate) exactly as drprg_hip_set_opts defaults them"""
    return (2 * k + 1, 0.001) if illumina else (250, 0.11)
