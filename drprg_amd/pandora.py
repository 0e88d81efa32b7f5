"""Host-side mirror of the reference's `Pandora` shim for the predict hot path.

`Pandora` below exposes the operator interface of struct Pandora in /root/reference/src/lib.rs:459-698
(`index_with`, `discover_with`, `genotype_with`, `vcf_filename`, `list_prgs_with_novel_variants`) with
the same argument meaning and error behaviour, but runs in-process on an MI355X through the C ABI of
include/drprg_hip.h instead of spawning the external `pandora` program.  `Context` is the thin object
wrapper of the C ABI that tests and bench.py drive directly.
"""
import ctypes as C
import os
import re

import numpy as np

from ._lib import MapOpts, lib

MTB_GENOME_SIZE = 4411532  # /root/reference/src/lib.rs:36


class DependencyError(RuntimeError):
    """Mirror of DependencyError (/root/reference/src/lib.rs:43-75): ProcessError / MissingExpectedOutput /
    NovelVariantParsingError are told apart by `kind`."""

    def __init__(self, kind, message, code=0):
        super().__init__(f"{kind}: {message}")
        self.kind = kind
        self.code = code


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib.drprg_hip_last_error(ctx)
        raise DependencyError("ProcessError", (msg or b"").decode() or f"error {rc}", code=-rc)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One opened index (+ its HIP device).  device=-1: host-only (index export, genotyping)."""

    def __init__(self, prg_file, w, k, device=0, from_files=True, threads=1, devices=None):
        """devices: a list of HIP device ids -> one context over all of them (drprg_hip_open_multi: map_fastx shards the reads)"""
        p = os.fsencode(prg_file)
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            self._h = lib.drprg_hip_open_multi(p, w, k, arr, len(devices), 1 if from_files else 0)
            device = devices[0]
        else:
            self._h = lib.drprg_hip_open(p, w, k, device) if from_files else lib.drprg_hip_open_prg(p, w, k, device, threads)
        if not self._h:
            raise DependencyError("ProcessError", lib.drprg_hip_last_error(None).decode())
        self.w, self.k, self.device = w, k, device
        sizes = (C.c_uint64 * 5)()
        lib.drprg_hip_index_sizes(self._h, sizes)
        self.n_keys, self.n_records, self.n_prgs, self.n_knodes, self.n_slots = (int(x) for x in sizes)

    def close(self):
        if self._h:
            lib.drprg_hip_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    # ---- options -------------------------------------------------------------------------------
    def set_opts(self, illumina=False, min_cluster_size=10, genome_size=MTB_GENOME_SIZE, max_diff=0, error_rate=0.0,
                 genotyping_error_rate=0.0, kernel=0, binomial=False):
        """kernel: 0 auto, 1 direct sketch kernel + generic cluster pipeline, 2 Bloom-prefiltered kernel, 3 direct sketch
        kernel in its candidate form (read_cluster_kernel)"""
        o = MapOpts(max_diff, error_rate, min_cluster_size, 1 if illumina else 0, genome_size, genotyping_error_rate, kernel,
                    1 if binomial else 0)
        _check(lib.drprg_hip_set_opts_sized(self._h, C.byref(o), C.sizeof(o)), self._h)

    # ---- mapping -------------------------------------------------------------------------------
    def set_threads(self, threads):
        """parser threads of map_fastx (the -t drprg forwards to pandora)"""
        _check(lib.drprg_hip_set_threads(self._h, int(threads)), self._h)

    def map_fastx(self, path):
        _check(lib.drprg_hip_map_fastx(self._h, os.fsencode(path)), self._h)

    def map_host(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(lib.drprg_hip_map_host(self._h, _ptr(bases), _ptr(offsets), len(offsets) - 1), self._h)

    # ---- 2-bit packed reads (include/drprg_hip.h "packed reads") -----------------------------------
    def set_input_format(self, packed):
        """map_fastx packs the reads to 2 bits on its parser threads (True) or hands ASCII blocks to the device (False, the default)"""
        _check(lib.drprg_hip_set_input_format(self._h, 1 if packed else 0), self._h)

    def map_host_packed(self, words, offsets, npos=None):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npos = np.ascontiguousarray(npos if npos is not None else [], dtype=np.uint64)
        _check(lib.drprg_hip_map_host_packed(self._h, _ptr(words), _ptr(offsets), len(offsets) - 1, _ptr(npos) if npos.size else None, npos.size), self._h)

    def map_device_packed(self, d_words, d_offsets, n_reads, n_bases, d_npos=None, n_npos=0, d_covg=None, d_prg_reads=None, stream=None, deferred=False):
        fn = lib.drprg_hip_map_device_packed_async if deferred else lib.drprg_hip_map_device_packed
        _check(fn(self._h, d_words, d_offsets, n_reads, n_bases, d_npos, n_npos, d_covg, d_prg_reads, stream), self._h)

    def pack_device(self, d_bases, n_bases, d_words, d_npos=None, npos_cap=0, stream=None):
        """ASCII -> packed on the device; returns the number of non-ACGT bases (their positions, ascending, in d_npos)"""
        n = C.c_uint64()
        _check(lib.drprg_hip_pack_device(self._h, d_bases, n_bases, d_words, d_npos, npos_cap, C.byref(n), stream), self._h)
        return int(n.value)

    def map_device(self, d_bases, d_offsets, n_reads, n_bases, d_covg=None, d_prg_reads=None, stream=None):
        """Pointers are integer device addresses (e.g. torch.Tensor.data_ptr())."""
        _check(lib.drprg_hip_map_device(self._h, d_bases, d_offsets, n_reads, n_bases, d_covg, d_prg_reads, stream), self._h)

    def map_device_async(self, d_bases, d_offsets, n_reads, n_bases, d_covg=None, d_prg_reads=None, stream=None):
        """map_device without the host waiting: the batch is queued and completed by the next call (or sync(), or anything
        that reads results); its buffers must stay valid and unchanged until then (drprg_hip_map_device_async)."""
        _check(lib.drprg_hip_map_device_async(self._h, d_bases, d_offsets, n_reads, n_bases, d_covg, d_prg_reads, stream), self._h)

    def sync(self):
        _check(lib.drprg_hip_sync(self._h), self._h)

    def reduce(self):
        """multi-device context: the devices' vectors summed into the first device, on the devices (drprg_hip_reduce)"""
        _check(lib.drprg_hip_reduce(self._h), self._h)
        buf = C.create_string_buffer(256)
        lib.drprg_hip_reduce_info(self._h, buf, len(buf))
        return buf.value.decode()

    # ---- coverage ------------------------------------------------------------------------------
    def coverage(self):
        covg = np.zeros(2 * self.n_knodes, dtype=np.uint32)
        prg_reads = np.zeros(self.n_prgs, dtype=np.uint32)
        _check(lib.drprg_hip_coverage(self._h, _ptr(covg), covg.size, _ptr(prg_reads), prg_reads.size), self._h)
        return covg, prg_reads

    def set_coverage(self, covg, prg_reads, total_bases):
        covg = np.ascontiguousarray(covg, dtype=np.uint32)
        prg_reads = np.ascontiguousarray(prg_reads, dtype=np.uint32)
        _check(lib.drprg_hip_set_coverage(self._h, _ptr(covg), covg.size, _ptr(prg_reads), prg_reads.size, total_bases), self._h)

    def device_coverage(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib.drprg_hip_device_coverage(self._h, C.byref(a), C.byref(b)), self._h)
        return a.value, b.value

    def reset(self):
        _check(lib.drprg_hip_reset(self._h), self._h)

    # ---- reads that stay in HBM (include/drprg_hip.h: drprg_hip_keep_reads) --------------------
    def keep_reads(self, max_bytes):
        """map_fastx leaves the blocks it copies on the device (up to max_bytes per device; 0 switches it off)"""
        _check(lib.drprg_hip_keep_reads(self._h, int(max_bytes)), self._h)

    def map_resident(self, other):
        """maps the reads `other` keeps in HBM against this context's index (DependencyError with code -61 if it does not hold them all)"""
        _check(lib.drprg_hip_map_resident(self._h, other._h), self._h)

    def resident_info(self):
        out = (C.c_uint64 * 4)()
        _check(lib.drprg_hip_resident_info(self._h, out), self._h)
        return dict(complete=bool(out[0]), bytes=int(out[1]), blocks=int(out[2]), last_discover_from_hbm=bool(out[3]))

    def counters(self):
        out = (C.c_uint64 * 8)()
        _check(lib.drprg_hip_counters(self._h, out), self._h)
        names = ("reads", "bases", "minimizers", "hits", "clusters_kept", "hits_kept", "kernel", "leftover_reads")
        return dict(zip(names, (int(x) for x in out)))

    # ---- genotyping ----------------------------------------------------------------------------
    def genotype(self, vcf_refs, out_vcf, sample="sample"):
        _check(lib.drprg_hip_genotype(self._h, os.fsencode(vcf_refs) if vcf_refs else None, os.fsencode(out_vcf),
                                      sample.encode()), self._h)
        gi = (C.c_uint32 * 4)()
        lib.drprg_hip_genotype_info(self._h, gi)
        return dict(exp_depth_covg=int(gi[0]), min_kmer_covg=int(gi[1]), loci_present=int(gi[2]), records=int(gi[3]))

    def coverage_model(self):
        """what estimate_parameters made of the coverage of the last genotype() (drprg_hip_coverage_model)"""
        out = (C.c_double * 12)()
        _check(lib.drprg_hip_coverage_model(self._h, out), self._h)
        names = ("exp_depth_covg", "binomial", "e_rate", "nb_p", "nb_r", "branch", "mean", "var", "reads_per_locus", "bin_p", "thresh",
                 "dropped_low_coverage")
        d = dict(zip(names, (float(x) for x in out)))
        for key in ("exp_depth_covg", "binomial", "branch", "reads_per_locus", "thresh", "dropped_low_coverage"):
            d[key] = int(d[key])
        return d

    def discover(self, vcf_refs, out_dir, sample="sample"):
        """candidate_regions.tsv + denovo_paths.txt ("0 loci": no local assembly) under out_dir; returns the regions"""
        n = C.c_uint32()
        _check(lib.drprg_hip_discover(self._h, os.fsencode(vcf_refs) if vcf_refs else None, os.fsencode(out_dir), sample.encode(),
                                      C.byref(n)), self._h)
        regions = []
        for line in open(os.path.join(out_dir, "candidate_regions.tsv")):
            if not line.startswith("#"):
                f = line.rstrip("\n").split("\t")
                regions.append(dict(locus=f[0], start=int(f[1]), end=int(f[2]), low_start=int(f[3]), low_end=int(f[4]),
                                    max_covg=int(f[5]), seq=f[6]))
        assert len(regions) == n.value
        return regions

    def discover_reads(self, reads, vcf_refs, out_dir, sample="sample", list_loci=True):
        """candidate regions + the pile-up of the reads over them; returns the novel variants [(locus, pos1, ref, alt, support, spanning)]"""
        out = (C.c_uint32 * 3)()
        _check(lib.drprg_hip_discover_reads(self._h, os.fsencode(reads), os.fsencode(vcf_refs) if vcf_refs else None, os.fsencode(out_dir),
                                            sample.encode(), 1 if list_loci else 0, out), self._h)
        variants = []
        for line in open(os.path.join(out_dir, "denovo_variants.tsv")):
            if not line.startswith("#"):
                f = line.rstrip("\n").split("\t")
                variants.append((f[0], int(f[1]), "" if f[2] == "." else f[2], "" if f[3] == "." else f[3], int(f[4]), int(f[5])))
        assert len(variants) == out[1]
        return variants

    def update_prg(self, out_prg):
        """the PRG file with the novel variants of the last discover_reads added as sites; returns how many were added"""
        n = C.c_uint32()
        _check(lib.drprg_hip_update_prg(self._h, os.fsencode(out_prg), C.byref(n)), self._h)
        return int(n.value)

    def update_prg_from_paths(self, denovo_paths, out_prg):
        """MakePrg::update (/root/reference/src/lib.rs:279-456) on a denovo_paths.txt -- this library's or pandora discover's own"""
        n = C.c_uint32()
        _check(lib.drprg_hip_update_prg_from_paths(self._h, os.fsencode(denovo_paths), os.fsencode(out_prg), C.byref(n)), self._h)
        return int(n.value)

    def save_coverage(self, path, tag):
        _check(lib.drprg_hip_save_coverage(self._h, os.fsencode(path), tag.encode()), self._h)

    def load_coverage(self, path, tag):
        """True if the cached vector belonged to this (PRG, reads, parameters) and is now installed"""
        rc = lib.drprg_hip_load_coverage(self._h, os.fsencode(path), tag.encode(), None)
        if rc == -2:
            return False
        _check(rc, self._h)
        return True

    def genotype_alleles(self, out_tsv):
        """per allele of the last genotype(): (chrom, pos, allele) -> global k-mer nodes its statistics were taken over"""
        _check(lib.drprg_hip_genotype_alleles(self._h, os.fsencode(out_tsv)), self._h)
        out = {}
        for line in open(out_tsv):
            if line.startswith("#"):
                continue
            chrom, pos, a, n, nodes = line.rstrip("\n").split("\t")
            out[(chrom, int(pos), int(a))] = np.array([int(x) for x in nodes.split(",") if x], dtype=np.int64)
        return out

    # ---- introspection -------------------------------------------------------------------------
    def filter_selfcheck(self):
        out = (C.c_uint64 * 8)()
        _check(lib.drprg_hip_filter_selfcheck(self._h, out), self._h)
        names = ("codes", "level0_false_negatives", "level12_false_negatives", "stage2_false_negatives", "level0_fill_permille",
                 "level12_fill_permille", "stage2_fill_permille", "shared_array_false_negatives")
        return dict(zip(names, (int(x) for x in out)))

    def table_tier(self):
        out = (C.c_uint64 * 6)()
        _check(lib.drprg_hip_device_tables(self._h, out), self._h)
        return dict(pbloom_words=int(out[0]), table_bytes=int(out[1]), lds_filter_bytes=int(out[2]), kernel=int(out[3]),
                    l2_filter_bytes=int(out[4]))

    def export_index(self):
        """Flat index (sorted keys + CSR records) in the layout oracle/oracle.c consumes."""
        keys = np.zeros(self.n_keys, dtype=np.uint64)
        rec_off = np.zeros(self.n_keys + 1, dtype=np.uint32)
        rec_prg = np.zeros(self.n_records, dtype=np.uint32)
        rec_knode = np.zeros(self.n_records, dtype=np.uint32)
        rec_strand = np.zeros(self.n_records, dtype=np.uint8)
        min_path_len = np.zeros(self.n_prgs, dtype=np.uint32)
        knode_base = np.zeros(self.n_prgs + 1, dtype=np.uint32)
        _check(lib.drprg_hip_index_export(self._h, _ptr(keys), _ptr(rec_off), _ptr(rec_prg), _ptr(rec_knode), _ptr(rec_strand),
                                          _ptr(min_path_len), _ptr(knode_base)), self._h)
        return dict(keys=keys, rec_off=rec_off, rec_prg=rec_prg, rec_knode=rec_knode, rec_strand=rec_strand,
                    min_path_len=min_path_len, knode_base=knode_base)

    def prg_nodes(self, prg):
        """(starts, ends, n_sites) of the local graph of PRG `prg`"""
        n, ns = C.c_uint32(), C.c_uint32()
        _check(lib.drprg_hip_prg_nodes(self._h, prg, None, None, 0, C.byref(n), C.byref(ns)), self._h)
        starts = np.zeros(n.value, dtype=np.uint32)
        ends = np.zeros(n.value, dtype=np.uint32)
        _check(lib.drprg_hip_prg_nodes(self._h, prg, _ptr(starts), _ptr(ends), n.value, C.byref(n), C.byref(ns)), self._h)
        return starts, ends, int(ns.value)

    def filter_schedule(self):
        """sketch_filter_kernel's chunk schedule in the batch completed last and when its wave classes were through (include/drprg_hip.h)"""
        out = (C.c_uint64 * 20)()
        _check(lib.drprg_hip_filter_schedule(self._h, out), self._h)
        v = [int(x) for x in out]
        return {"form": "dynamic" if v[0] > 1 else "static", "rounds": v[0], "slices": v[1], "slices_per_workgroup": v[11], "round0_tiles_per_wave": v[2],
                "round0_class_shares_per_256": v[3:7], "chunk_tiles_per_round": [x for x in v[13:20] if x],
                "class_end_us": [round(x / 100.0, 1) for x in v[7:11]]}

    def kernel_timing(self, enable=True, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        _check(lib.drprg_hip_kernel_timing(self._h, 1 if enable else 0, 1 if reset else 0, C.byref(ms), C.byref(n)), self._h)
        return ms.value, int(n.value)


def allele_stats(fwd, rev, min_kmer_covg):
    """the product's per-allele statistics on raw k-mer coverages: ([MEAN_FWD, MEAN_REV, MED_FWD, MED_REV, SUM_FWD, SUM_REV], GAPS)"""
    fwd = np.ascontiguousarray(fwd, dtype=np.uint32)
    rev = np.ascontiguousarray(rev, dtype=np.uint32)
    out = np.zeros(6, np.uint32)
    gaps = C.c_double()
    _check(lib.drprg_hip_allele_stats(_ptr(fwd), _ptr(rev), len(fwd), min_kmer_covg, _ptr(out), C.byref(gaps)))
    return [int(x) for x in out], gaps.value


def genotype_site(mean_fwd, mean_rev, gaps, e, eps=0.01):
    """the product's likelihood / GT / GT_CONF of one site from its per-allele means and gaps"""
    mf = np.ascontiguousarray(mean_fwd, dtype=np.uint32)
    mr = np.ascontiguousarray(mean_rev, dtype=np.uint32)
    g = np.ascontiguousarray(gaps, dtype=np.float64)
    lik = np.zeros(len(mf), np.float64)
    gt, conf = C.c_int32(), C.c_double()
    _check(lib.drprg_hip_genotype_site(_ptr(mf), _ptr(mr), _ptr(g), len(mf), e, eps, _ptr(lik), C.byref(gt), C.byref(conf)))
    return lik, gt.value, conf.value


class Pandora:
    """In-process stand-in for struct Pandora (/root/reference/src/lib.rs:459-698)."""

    def __init__(self, device=0):
        self.device = device

    @staticmethod
    def _parse_args(args):
        """The argv drprg passes: -t T -w W -k K -c C [-I] [-K] (/root/reference/src/predict.rs:236-245)."""
        o = dict(threads=1, w=14, k=15, c=10, illumina=False)
        it = iter([str(a) for a in args])
        for a in it:
            if a == "-t":
                o["threads"] = int(next(it))
            elif a == "-w":
                o["w"] = int(next(it))
            elif a == "-k":
                o["k"] = int(next(it))
            elif a == "-c":
                o["c"] = int(next(it))
            elif a == "-I":
                o["illumina"] = True
            elif a == "-K":
                pass
            else:
                raise DependencyError("ProcessError", f"unknown pandora option {a}", code=2)
        return o

    def index_with(self, prg_path, args=()):
        """Pandora::index_with, /root/reference/src/lib.rs:479-510."""
        o = self._parse_args(args)
        _check(lib.drprg_hip_index(os.fsencode(prg_path), o["w"], o["k"], o["threads"]))

    def _mapped_context(self, prg, reads, args):
        o = self._parse_args(args)
        ctx = Context(prg, o["w"], o["k"], device=self.device)
        ctx.set_opts(illumina=o["illumina"], min_cluster_size=o["c"], genome_size=MTB_GENOME_SIZE)
        ctx.map_fastx(reads)
        return ctx

    @staticmethod
    def _run_tag(prg, reads, args):
        """identifies (PRG file, reads file, mapping parameters): what a cached coverage vector belongs to"""
        def stamp(p):
            st = os.stat(p)
            return f"{os.path.realpath(p)}:{st.st_size}:{st.st_mtime_ns}"
        return "|".join([stamp(prg), stamp(reads)] + [str(a) for a in args if str(a) != "-K"])

    COVERAGE_CACHE = ".drprg_hip_coverage"

    def discover_with(self, prg, query_idx, outdir, args=(), list_loci=False):
        """Pandora::discover_with, /root/reference/src/lib.rs:513-578.  Returns the denovo_paths.txt path.

        The mapping half runs (its coverage vector is kept under `outdir` for the genotype_with call that follows), candidate
        regions go to candidate_regions.tsv, and the reads are piled up over them: novel variants go to
        denovo_variants.tsv.  list_loci=True also lists them in denovo_paths.txt (the caller's make_prg update then runs on it;
        the layout follows the one example in the reference tree and could not be tried on make_prg here); the default keeps
        "0 loci with denovo variants"."""
        import warnings
        os.makedirs(outdir, exist_ok=True)
        with open(query_idx) as fh:
            sample, reads = fh.readline().split()[:2]
        with self._mapped_context(prg, reads, args) as ctx:
            ctx.set_threads(self._parse_args(args)["threads"])
            self.novel_variants = ctx.discover_reads(reads, None, outdir, sample, list_loci=list_loci)
            ctx.save_coverage(os.path.join(outdir, self.COVERAGE_CACHE), self._run_tag(prg, reads, args))
        if self.novel_variants and not list_loci:
            warnings.warn(f"discover: {len(self.novel_variants)} novel variant(s) found (denovo_variants.tsv) but not listed in "
                          "denovo_paths.txt: the PRG will not be updated")
        path = os.path.join(outdir, "denovo_paths.txt")
        if not os.path.exists(path):
            raise DependencyError("MissingExpectedOutput", path)
        return path

    def genotype_with(self, prg, vcf_ref, reads, outdir, args=()):
        """Pandora::genotype_with, /root/reference/src/lib.rs:580-642.  If discover_with left this run's coverage vector
        under <outdir>/discover (where drprg puts it, /root/reference/src/predict.rs:248), the reads are not mapped again."""
        os.makedirs(outdir, exist_ok=True)
        o = self._parse_args(args)
        cache = os.path.join(outdir, "discover", self.COVERAGE_CACHE)
        ctx = Context(prg, o["w"], o["k"], device=self.device)
        with ctx:
            ctx.set_opts(illumina=o["illumina"], min_cluster_size=o["c"], genome_size=MTB_GENOME_SIZE)
            self.reused_discover_coverage = os.path.exists(cache) and ctx.load_coverage(cache, self._run_tag(prg, reads, args))
            if not self.reused_discover_coverage:
                ctx.map_fastx(reads)
            ctx.genotype(vcf_ref, os.path.join(outdir, self.vcf_filename()))

    @staticmethod
    def vcf_filename():
        """/root/reference/src/lib.rs:644-646"""
        return "pandora_genotyped.vcf"

    @staticmethod
    def list_prgs_with_novel_variants(denovo_file):
        """/root/reference/src/lib.rs:648-697"""
        try:
            contents = open(denovo_file).read()
        except OSError:
            raise DependencyError("NovelVariantParsingError", f"Unable to read {denovo_file!r}")
        m = re.search(r"\n(?P<num>\d+) loci with denovo variants\n", contents)
        if not m:
            raise DependencyError("NovelVariantParsingError", "Unable to find line describing the number of novel variants")
        expected = int(m.group("num"))
        genes, prev = [], ""
        for line in contents.splitlines():
            if line.endswith("nodes"):
                genes.append(prev)
            prev = line
        if len(genes) != expected:
            raise DependencyError("NovelVariantParsingError",
                                  f"Expected {expected} genes with novel variants, but found {len(genes)}")
        return genes


def pack_reads(bases):
    """host helper: uint8 ASCII bases -> (uint32 words, uint64 ascending positions of the bases that are not ACGTacgt)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    words = np.zeros((bases.size + 15) // 16, dtype=np.uint32)
    cap = 1024
    while True:
        npos = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint64()
        rc = lib.drprg_hip_pack_reads(_ptr(bases), bases.size, _ptr(words), _ptr(npos), cap, C.byref(n))
        if rc == 0:
            return words, npos[:n.value].copy()
        if rc != -75:  # -EOVERFLOW: n holds the number needed
            _check(rc)
        cap = int(n.value)
