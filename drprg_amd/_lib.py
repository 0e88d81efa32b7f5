"""ctypes binding of libdrprg_hip.so (the C ABI of include/drprg_hip.h).

There is no Python or CPU fallback for the hot path: if the shared library has not been built this
module raises ImportError with the build command, and every map call on a GPU-less host returns
-ENODEV from the library itself.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRPRG_HIP_LIB") or os.path.join(_HERE, "lib", "libdrprg_hip.so")  # (override: sanitizer builds)
PANDORA_EXE = os.path.join(_HERE, "bin", "pandora")


class MapOpts(C.Structure):
    """struct drprg_hip_map_opts"""
    _fields_ = [
        ("max_diff", C.c_int32),
        ("error_rate", C.c_double),
        ("min_cluster_size", C.c_uint32),
        ("illumina", C.c_int32),
        ("genome_size", C.c_uint64),
        ("genotyping_error_rate", C.c_double),
        ("kernel", C.c_int32),
        ("binomial", C.c_int32),
    ]


class AnnotateOpts(C.Structure):
    """struct drprg_hip_annotate_opts; defaults = the CLI defaults of `drprg predict`
    (/root/reference/src/filter.rs:12-16, src/minor.rs:11-17)"""
    _fields_ = [
        ("min_covg", C.c_int32), ("max_covg", C.c_int32),
        ("min_strand_bias", C.c_float), ("min_gt_conf", C.c_float), ("min_frs", C.c_float),
        ("max_indel", C.c_int32),
        ("maf", C.c_float), ("max_gaps", C.c_float), ("max_called_gaps", C.c_float), ("max_gaps_diff", C.c_float),
        ("minor_min_covg", C.c_int32), ("minor_min_strand_bias", C.c_float),
        ("ignore_synonymous", C.c_int32), ("id_seed", C.c_uint64),
    ]

    @classmethod
    def cli_defaults(cls, illumina=False):
        return cls(3, 2 ** 31 - 1, 0.01, 0.0, 0.0, -1, 0.1 if illumina else 1.0, 0.5, 0.39, 0.2, 3, 0.01, 0, 0)


# name -> (restype, argtypes); every symbol include/drprg_hip.h declares
SIGNATURES = {
    "drprg_hip_index": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "drprg_hip_open": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "drprg_hip_open_prg": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "drprg_hip_open_multi": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]),
    "drprg_hip_reduce": (C.c_int, [C.c_void_p]),
    "drprg_hip_reduce_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "drprg_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "drprg_hip_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "drprg_hip_comm_destroy": (C.c_int, [C.c_void_p]),
    "drprg_hip_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "drprg_hip_close": (None, [C.c_void_p]),
    "drprg_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "drprg_hip_experimental": (C.c_int, []),
    "drprg_hip_set_opts": (C.c_int, [C.c_void_p, C.POINTER(MapOpts)]),
    "drprg_hip_set_opts_sized": (C.c_int, [C.c_void_p, C.POINTER(MapOpts), C.c_size_t]),
    "drprg_hip_map_fastx": (C.c_int, [C.c_void_p, C.c_char_p]),
    "drprg_hip_set_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "drprg_hip_parse_fastx": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "drprg_hip_gunzip_file": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "drprg_hip_map_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "drprg_hip_map_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "drprg_hip_map_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "drprg_hip_sync": (C.c_int, [C.c_void_p]),
    "drprg_hip_set_input_format": (C.c_int, [C.c_void_p, C.c_int]),
    "drprg_hip_pack_reads": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "drprg_hip_map_host_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "drprg_hip_map_device_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "drprg_hip_map_device_packed_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]),
    "drprg_hip_pack_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]),
    "drprg_hip_coverage_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "drprg_hip_coverage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "drprg_hip_set_coverage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]),
    "drprg_hip_device_coverage": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "drprg_hip_reset": (C.c_int, [C.c_void_p]),
    "drprg_hip_keep_reads": (C.c_int, [C.c_void_p, C.c_uint64]),
    "drprg_hip_map_resident": (C.c_int, [C.c_void_p, C.c_void_p]),
    "drprg_hip_resident_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "drprg_hip_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "drprg_hip_genotype": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "drprg_hip_genotype_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "drprg_hip_discover": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "drprg_hip_discover_reads": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint32)]),
    "drprg_hip_update_prg": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "drprg_hip_update_prg_from_paths": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "drprg_hip_save_coverage": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "drprg_hip_load_coverage": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]),
    "drprg_hip_allele_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]),
    "drprg_hip_genotype_site": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "drprg_hip_genotype_alleles": (C.c_int, [C.c_void_p, C.c_char_p]),
    "drprg_hip_estimate_parameters": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_int,
                                               C.POINTER(C.c_double)]),
    "drprg_hip_kmer_log_prob": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "drprg_hip_prob_threshold": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]),
    "drprg_hip_max_path": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "drprg_hip_path_base_coverage": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                              C.POINTER(C.c_uint64)]),
    "drprg_hip_path_coverage_too_low": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "drprg_hip_coverage_model": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "drprg_hip_index_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "drprg_hip_filter_selfcheck": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "drprg_hip_device_tables": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "drprg_hip_index_export": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "drprg_hip_prg_nodes": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]),
    "drprg_hip_annotate": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(AnnotateOpts), C.c_char_p, C.c_size_t]),
    "drprg_hip_vcf_to_bcf": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "drprg_hip_report_json": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p,
                                        C.c_size_t]),
    "drprg_hip_kernel_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "drprg_hip_filter_schedule": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (`make` at the repo root or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.  If this library pulls in
    # /opt/rocm's copy first and torch is imported afterwards, the process holds two runtimes and the second one sees no
    # device.  Importing torch first makes our DT_NEEDED resolve to the copy that is already loaded.  (Plumbing only:
    # nothing here uses torch; without torch installed the system runtime is the only one anyway.)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
