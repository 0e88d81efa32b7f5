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
        # oracle_index.c
        L.orc_index_prg.restype = C.c_void_p
        L.orc_index_prg.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.orc_kg_free.restype = None
        L.orc_kg_free.argtypes = [C.c_void_p]
        for name in ("orc_kg_n_nodes", "orc_kg_min_path_len", "orc_kg_n_local_nodes"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_kg_n_edges.restype = C.c_uint64
        L.orc_kg_n_edges.argtypes = [C.c_void_p]
        L.orc_kg_local_nodes.restype = None
        L.orc_kg_local_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_kg_kmers.restype = None
        L.orc_kg_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_kg_edges.restype = None
        L.orc_kg_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_kg_path.restype = C.c_uint32
        L.orc_kg_path.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_kg_walk_check.restype = C.c_int
        L.orc_kg_walk_check.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        # oracle_vcf.c
        L.orc_vcf_sites.restype = C.c_void_p
        L.orc_vcf_sites.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_free.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_vcf_set_overlap_rule.restype = None
        L.orc_vcf_set_overlap_rule.argtypes = [C.c_int]
        L.orc_set_hash_mode.restype = None
        L.orc_set_hash_mode.argtypes = [C.c_int, C.c_uint64]

    # ---- a-9, front half: the oracle's own VCF sites and allele -> k-mer nodes (oracle/oracle_vcf.c) ---------------
    def set_hash_mode(self, mode, seed=0):
        """0 = the hash of the path (minimap hash64); 1.. = control hashes (tests/test_kmer_count_kat.py only)"""
        self.lib.orc_set_hash_mode(int(mode), int(seed))

    def vcf_sites(self, prg_string, w, k, refseq=None):
        """records of one locus: [dict(pos, ref, alts, vc, graphtype, knodes=[per allele: sorted local k-mer node ids])],
        plus (threaded, [local nodes of the reference walk])"""
        L = self.lib
        g = L.orc_index_prg(prg_string.encode() if isinstance(prg_string, str) else prg_string, w, k)
        if not g:
            raise ValueError("malformed PRG string")
        try:
            p = L.orc_vcf_sites(g, refseq.encode() if isinstance(refseq, str) else refseq)
            if not p:
                raise ValueError("inconsistent local graph")
            text = C.string_at(p).decode()
            L.orc_free(p)
        finally:
            L.orc_kg_free(g)
        recs, refpath = [], None
        for line in text.splitlines():
            t = line.split("\t")
            if t[0] == "#refpath":
                refpath = (t[1] == "1", [int(x) for x in t[2].split(",")] if t[2] else [])
                continue
            key = (int(t[0]), t[1], t[2])
            if not recs or recs[-1]["key"] != key:
                recs.append(dict(key=key, pos=int(t[0]), ref=t[1], alts=t[2].split(","), vc=t[3], graphtype=t[4], knodes=[]))
            assert int(t[5]) == len(recs[-1]["knodes"])
            ids = [int(x) for x in t[7].split(",")] if len(t) > 7 and t[7] else []
            assert len(ids) == int(t[6])
            recs[-1]["knodes"].append(ids)
        return recs, refpath

    # ---- a-4: the oracle's own index (oracle/oracle_index.c); nothing here comes from the product ----------------
    def sketch_prg(self, prg_string, w, k, paths=False, walk_check=False):
        """k-mer graph of one PRG string: dict(n_nodes, min_path_len, hash, strand, edges[, paths][, walk])"""
        L = self.lib
        g = L.orc_index_prg(prg_string.encode() if isinstance(prg_string, str) else prg_string, w, k)
        if not g:
            raise ValueError("malformed PRG string")
        try:
            n = L.orc_kg_n_nodes(g)
            h = np.zeros(n - 2, np.uint64)
            st = np.zeros(n - 2, np.uint8)
            L.orc_kg_kmers(g, _p(h), _p(st))
            ne = L.orc_kg_n_edges(g)
            ef = np.zeros(ne, np.uint32)
            et = np.zeros(ne, np.uint32)
            L.orc_kg_edges(g, _p(ef), _p(et))
            nl = L.orc_kg_n_local_nodes(g)
            ls = np.zeros(nl, np.uint32)
            le = np.zeros(nl, np.uint32)
            L.orc_kg_local_nodes(g, _p(ls), _p(le))
            out = dict(n_nodes=int(n), min_path_len=int(L.orc_kg_min_path_len(g)), hash=h, strand=st,
                       edges=np.stack([ef, et], axis=1), local_starts=ls, local_ends=le)
            if paths:
                iv = np.zeros(2 * 128, np.uint32)
                out["paths"] = []
                for i in range(1, n - 1):
                    m = L.orc_kg_path(g, i, _p(iv), 128)
                    out["paths"].append([(int(iv[2 * j]), int(iv[2 * j + 1])) for j in range(m)])
            if walk_check:
                res = np.zeros(4, np.uint64)
                conf = np.zeros(max(n - 2, 1), np.uint8)
                rc = L.orc_kg_walk_check(g, _p(res), _p(conf))
                assert rc == 0
                out["walk"] = dict(windows=int(res[0]), missing=int(res[1]), confirmed=int(res[2]), unconfirmed=int(res[3]),
                                   mask=conf[:n - 2].astype(bool))
            return out
        finally:
            L.orc_kg_free(g)

    def build_index(self, prg_strings, w, k):
        """The flat index oracle.c's orc_map_reads consumes, built by the oracle itself from PRG strings (same layout as
        Context.export_index(): sorted distinct keys, CSR records ordered by (key, prg, k-mer node), global node numbers)."""
        graphs = [self.sketch_prg(s, w, k) for s in prg_strings]
        knode_base = np.zeros(len(graphs) + 1, np.uint32)
        knode_base[1:] = np.cumsum([g["n_nodes"] for g in graphs])
        key = np.concatenate([g["hash"] for g in graphs]) if graphs else np.zeros(0, np.uint64)
        prg = np.concatenate([np.full(len(g["hash"]), i, np.uint32) for i, g in enumerate(graphs)])
        knode = np.concatenate([np.arange(1, len(g["hash"]) + 1, dtype=np.uint32) for g in graphs])
        strand = np.concatenate([g["strand"] for g in graphs])
        order = np.lexsort((knode, prg, key))
        key, prg, knode, strand = key[order], prg[order], knode[order], strand[order]
        keys, first = np.unique(key, return_index=True)
        rec_off = np.concatenate([first, [len(key)]]).astype(np.uint32)
        return dict(keys=keys.astype(np.uint64), rec_off=rec_off, rec_prg=prg.astype(np.uint32),
                    rec_knode=(knode_base[prg] + knode).astype(np.uint32), rec_strand=strand.astype(np.uint8),
                    min_path_len=np.array([g["min_path_len"] for g in graphs], np.uint32), knode_base=knode_base)

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


# ---- PRGs made of a fixture VCF's own sites (tests/test_kmer_count_kat.py, tests/test_vcf_sites.py) ------------------------------
def read_fasta_dict(path):
    out, name = {}, None
    for line in open(path):
        if line.startswith(">"):
            name = line[1:].split()[0]
            out[name] = ""
        elif name:
            out[name] += line.strip().upper()
    return out


def flat_prgs_from_sites(genes, records):
    """One PRG string per gene of `genes` (name -> sequence): the gene's sequence with every record (dicts with chrom, pos (1-based),
    ref, alts) that fits turned into a flat site REF | ALT1 | ALT2 ... in make_prg syntax.  A record whose alleles all start with the
    same base and one of which is that base alone is a padded indel (pandora pads an empty allele with the reference base in front
    of the site): the site is put back behind that base.  Returns (names, prg strings, keys (chrom, pos) of the records that were
    placed, as (chrom, pos, REF as printed)); a record that overlaps one already placed (a nested site of the real PRG) or whose REF is not the gene's sequence at POS
    is left out."""
    from drprg_amd import synth
    by = {}
    for r in records:
        by.setdefault(r["chrom"], []).append(r)
    names, prgs, placed = [], [], set()
    for gname, gseq in genes.items():
        segs, cur = [], 0
        for r in sorted(by.get(gname, []), key=lambda r: (r["pos"], -len(r["ref"]))):
            p0, ref, alts = r["pos"] - 1, r["ref"], list(r["alts"])
            if gseq[p0:p0 + len(ref)] != ref:
                continue
            if all(a[:1] == ref[:1] for a in alts) and min(len(x) for x in alts + [ref]) == 1:
                p0, ref, alts = p0 + 1, ref[1:], [a[1:] for a in alts]
            if p0 < cur:
                continue
            segs += [gseq[cur:p0], synth.Site([[ref]] + [[a] for a in alts])]
            cur = p0 + len(ref)
            placed.add((gname, r["pos"], r["ref"]))
        segs.append(gseq[cur:])
        names.append(gname)
        prgs.append(synth.prg_string(segs))
    return names, prgs, placed


def product_sites(names, prgs, refs, w, k, tmpdir):
    """The product's records and per-allele k-mer nodes for PRG strings (host-only context, every locus present):
    {name: [dict(pos, ref, alts, vc, graphtype, knodes=[per allele: sorted LOCAL k-mer node ids])]} in VCF order."""
    from drprg_amd import Context
    prg, genes = os.path.join(str(tmpdir), "dr.prg"), os.path.join(str(tmpdir), "genes.fa")
    with open(prg, "w") as fh:
        for n, s in zip(names, prgs):
            fh.write(f">{n}\n{s}\n")
    if refs is not None:
        with open(genes, "w") as fh:
            for n, s in zip(names, refs):
                fh.write(f">{n}\n{s}\n")
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=4411532)
    ctx.set_coverage(np.ones(2 * ctx.n_knodes, np.uint32), np.full(ctx.n_prgs, 100, np.uint32), 1000)  # every locus has clusters, none is dropped
    vcf, tsv = os.path.join(str(tmpdir), "o.vcf"), os.path.join(str(tmpdir), "alleles.tsv")
    ctx.genotype(genes if refs is not None else None, vcf)
    ctx.genotype_alleles(tsv)
    base = {n: int(b) for n, b in zip(names, ctx.export_index()["knode_base"])}
    rows = iter([line.rstrip("\n").split("\t") for line in open(tsv) if not line.startswith("#")])  # (in the order of the VCF's records)
    out = {n: [] for n in names}
    for line in open(vcf):
        if line.startswith("#"):
            continue
        t = line.rstrip("\n").split("\t")
        info = dict(x.split("=") for x in t[7].split(";"))
        rec = dict(pos=int(t[1]), ref=t[3], alts=t[4].split(","), vc=info["VC"], graphtype=info["GRAPHTYPE"], knodes=[])
        for a in range(1 + len(rec["alts"])):
            row = next(rows)
            assert (row[0], int(row[1]), int(row[2])) == (t[0], rec["pos"], a)
            rec["knodes"].append(sorted(int(x) - base[t[0]] for x in row[4].split(",") if x))
        out[t[0]].append(rec)
    ctx.close()
    return out


# ---- the oracle's whole pandora_genotyped.vcf: oracle_params.c (model, best path, presence) + oracle_vcf.c (records, allele -> k-mer
# nodes) + oracle.c (statistics, likelihood) + the text layout of the reference's fixture -----------------------------------------
def oracle_vcf_text(oracle, names, prgs, refs, covg, prg_reads, total_bases, w, k, genome_size, error_rate, binomial=False, eps=0.01,
                    sample="sample"):
    """What drprg_hip_genotype writes (minus the ##fileDate line), from the oracle's own pieces.  refs: {name: sequence} or None.
    Returns (text, dict(e, min_kmer_covg, present, dropped))."""
    L = oracle.lib
    L.orc_estimate_parameters.restype = None
    L.orc_estimate_parameters.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_int, C.c_void_p]
    L.orc_kmer_log_prob.restype = C.c_float
    L.orc_kmer_log_prob.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_prob_threshold.restype = C.c_int
    L.orc_prob_threshold.argtypes = [C.c_void_p, C.c_int64]
    L.orc_max_path.restype = C.c_int64
    L.orc_max_path.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int64]
    L.orc_path_coverage_too_low.restype = C.c_int
    L.orc_path_coverage_too_low.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
    L.orc_kg_base_coverage.restype = C.c_int64
    L.orc_kg_base_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    covg = np.minimum(np.asarray(covg, np.uint32), 65535).astype(np.uint32)  # pandora's u16 counters
    graphs = [L.orc_index_prg(s.encode(), w, k) for s in prgs]
    try:
        assert all(graphs)
        n_nodes = [int(L.orc_kg_n_nodes(g)) for g in graphs]
        base = np.concatenate([[0], np.cumsum(n_nodes)]).astype(np.int64)
        assert covg.size == 2 * base[-1]
        fwd = [covg[2 * base[i]:2 * base[i + 1]:2] for i in range(len(prgs))]
        rev = [covg[2 * base[i] + 1:2 * base[i + 1]:2] for i in range(len(prgs))]
        with_reads = [i for i in range(len(prgs)) if prg_reads[i] > 0]
        kc = np.concatenate([(fwd[i] + rev[i])[1:-1] for i in with_reads]).astype(np.uint32) if with_reads else np.zeros(0, np.uint32)
        gcov = int(min(int(total_bases) // max(int(genome_size), 1), 0xFFFFFFFF))
        a = np.zeros(10)
        L.orc_estimate_parameters(_p(np.ascontiguousarray(kc)), len(kc), int(sum(int(prg_reads[i]) for i in with_reads)), len(with_reads), gcov, k,
                                  error_rate, 1 if binomial else 0, _p(a))
        e = int(a[0])

        def logp(i):
            out = np.zeros(n_nodes[i], np.float32)
            for j in range(1, n_nodes[i] - 1):
                out[j] = L.orc_kmer_log_prob(int(a[1]), a[3], a[4], a[9], int(fwd[i][j]), int(rev[i][j]), int(prg_reads[i]))
            return out

        lps = {i: logp(i) for i in with_reads}
        all_lp = np.ascontiguousarray(np.concatenate([lps[i][1:-1] for i in with_reads]) if with_reads else np.zeros(0, np.float32), np.float32)
        thresh = L.orc_prob_threshold(_p(all_lp), len(all_lp))
        present, dropped = [], []
        for i in with_reads:
            n = n_nodes[i]
            ne = int(L.orc_kg_n_edges(graphs[i]))
            ef, et = np.zeros(ne, np.uint32), np.zeros(ne, np.uint32)
            L.orc_kg_edges(graphs[i], _p(ef), _p(et))
            path = np.zeros(n, np.uint32)
            m = L.orc_max_path(n, ne, _p(ef), _p(et), _p(lps[i]), thresh, 100, _p(path), n)
            if m == 0:
                continue
            total = np.ascontiguousarray(fwd[i] + rev[i], np.uint32)
            cap = 1 << 20
            ob = np.zeros(cap, np.uint32)
            nb = L.orc_kg_base_coverage(graphs[i], _p(path), m, _p(total), _p(ob), cap)
            assert nb <= cap
            if L.orc_path_coverage_too_low(_p(ob), nb, gcov):
                dropped.append(names[i])
                continue
            present.append(i)
    finally:
        for g in graphs:
            if g:
                L.orc_kg_free(g)
    thr = e // 10
    recs = []
    for i in present:
        for r in oracle.vcf_sites(prgs[i], w, k, refs.get(names[i]) if refs else None)[0]:
            stats, gaps = [], []
            for ids in r["knodes"]:
                idx = np.asarray(ids, np.int64)
                st, g = oracle.allele_stats(fwd[i][idx], rev[i][idx], thr)
                stats.append(st)
                gaps.append(g)
            lik, gt, conf = oracle.genotype([s[0] for s in stats], [s[1] for s in stats], gaps, float(e), eps)
            recs.append((names[i], r["pos"], r["ref"], r["alts"], r["vc"], r["graphtype"], stats, gaps, lik, gt, conf))
    recs.sort(key=lambda x: (x[0], x[1], x[2], x[3]))
    head = []
    for line in open(os.path.join(GOLDEN, "pandora_vcf_surface", "header.vcf")):
        if line.startswith("##contig") or line.startswith("#CHROM"):
            break
        if not line.startswith("##fileDate") and not line.startswith("##FILTER=<ID=PASS"):  # (the PASS line is bcftools', from when the fixture was cut down)
            head.append(line)
    out = head + [f"##contig=<ID={n}>\n" for n in sorted(names[i] for i in present)]
    out.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample + "\n")
    g6 = lambda v: "%g" % v
    for chrom, pos, ref, alts, vc, gtype, stats, gaps, lik, gt, conf in recs:
        cols = [",".join(str(s[j]) for s in stats) for j in range(6)]
        out.append(f"{chrom}\t{pos}\t.\t{ref}\t{','.join(alts)}\t.\t.\tVC={vc};GRAPHTYPE={gtype}\t"
                   "GT:MEAN_FWD_COVG:MEAN_REV_COVG:MED_FWD_COVG:MED_REV_COVG:SUM_FWD_COVG:SUM_REV_COVG:GAPS:LIKELIHOOD:GT_CONF\t"
                   f"{gt}:{':'.join(cols)}:{','.join(g6(x) for x in gaps)}:{','.join(g6(x) for x in lik)}:{g6(conf)}\n")
    return "".join(out), dict(e=e, min_kmer_covg=thr, present=sorted(names[i] for i in present), dropped=sorted(dropped))


def vcf_without_date(path):
    return "".join(line for line in open(path) if not line.startswith("##fileDate"))


# ---- discover: the oracle's two pile-ups (oracle/oracle_denovo.py) on the candidate regions a discover run wrote -------------------
def oracle_local_assembly(discover_dir, consensus, bases, offs):
    """[(locus, pos0, ref, alt, support, spanning)] by the oracle's own statement of the local (de Bruijn) assembly of every candidate region
    in <discover_dir>/candidate_regions.tsv: region order, the alleles of a region best supported first"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_denovo", os.path.join(ROOT, "oracle", "oracle_denovo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    regions = []
    for line in open(os.path.join(str(discover_dir), "candidate_regions.tsv")):
        if not line.startswith("#"):
            f = line.split("\t")
            regions.append((f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4])))
    text = np.asarray(bases, np.uint8).tobytes().decode()
    reads = [text[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    return mod.local_assembly(consensus, regions, reads)


def oracle_denovo(discover_dir, consensus, bases, offs, noisy=False):
    """[(locus, pos0, ref, alt, support, spanning)] by the oracle's separate statement of the accurate-read pile-up (noisy = False) or of the
    noisy-read column vote, over the regions in <discover_dir>/candidate_regions.tsv and the consensus {locus: sequence} the test knows"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_denovo", os.path.join(ROOT, "oracle", "oracle_denovo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    regions = []
    for line in open(os.path.join(str(discover_dir), "candidate_regions.tsv")):
        if not line.startswith("#"):
            f = line.split("\t")
            regions.append((f[0], int(f[1]), int(f[2])))
    text = np.asarray(bases, np.uint8).tobytes().decode()
    reads = [text[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    return (mod.column_vote if noisy else mod.pile_up)(consensus, regions, reads)


def prg_tree(prg):
    """A PRG string as a tree -- [plain sequence | [allele, allele, ...]] with every allele such a list again --, parsed the way pandora's
    LocalPRG::build_graph does [UPSTREAM-MEMORY]: an interval that holds a marker is split at the NEXT site number (5, 7, 9 ... in the order
    the parser meets the sites: a site before the sites inside it, then the sites behind it), what stands before that site must be plain
    sequence, the alleles are parsed in turn, then what stands behind the site.  Raises AssertionError on a string pandora would refuse
    (markers out of that order, unbalanced sites)."""
    import re
    next_site = [5]

    def build(s):
        if not re.search(r"\d", s):
            assert re.fullmatch(r"[ACGT ]*", s), s
            return [s.replace(" ", "")]
        m = next_site[0]
        tok, sep = " %d " % m, " %d " % (m + 1)
        i = s.find(tok)
        assert i >= 0, "site %d is not the next site of %r" % (m, s[:60])
        first = s[:i]
        assert not re.search(r"\d", first), "a site before site %d in %r" % (m, s[:60])
        j = s.find(tok, i + len(tok) - 1)  # (the closing marker may share the space of an empty last allele: " 6  5 ")
        assert j >= 0, "site %d does not close" % m
        middle, rest = s[i + len(tok):j] if j >= i + len(tok) else "", s[j + len(tok):]
        next_site[0] += 2
        alleles = middle.split(sep)
        assert len(alleles) >= 2, "site %d has one allele" % m
        return [first.replace(" ", ""), [build(a) for a in alleles]] + build(rest)

    return build(prg)


def prg_language(prg):
    """Every sequence a PRG string spells (prg_tree's rules).  Small PRGs only: the number of sequences is the product over the sites."""
    def spell(items):
        out = {""}
        for it in items:
            if isinstance(it, str):
                out = {x + it for x in out}
            else:
                alts = set()
                for allele in it:
                    alts |= spell(allele)
                out = {x + a for x in out for a in alts}
        return out

    return spell(prg_tree(prg))


def prg_spells(prg, seq):
    """does the PRG string (prg_tree's rules) spell `seq`?  Sets of reachable positions: fine for PRGs of any size"""
    def step(items, starts):
        for it in items:
            if not starts:
                return starts
            if isinstance(it, str):
                starts = {p + len(it) for p in starts if seq.startswith(it, p)}
            else:
                ends = set()
                for allele in it:
                    ends |= step(allele, starts)
                starts = ends
        return starts

    return len(seq) in step(prg_tree(prg), {0})


def prg_walks(prg, n, seed=0):
    """n random sequences of the PRG string (a random allele at every site)"""
    rng = np.random.default_rng(seed)

    def walk(items):
        return "".join(it if isinstance(it, str) else walk(it[int(rng.integers(len(it)))]) for it in items)

    tree = prg_tree(prg)
    return [walk(tree) for _ in range(n)]
