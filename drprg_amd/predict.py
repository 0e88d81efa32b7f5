"""Host-side mirror of the post-VCF half of `drprg predict` (/root/reference/src/predict.rs:420-1139) over the C ABI."""
import ctypes as C
import os

from ._lib import AnnotateOpts, lib
from .pandora import DependencyError


def _call(fn, *args):
    err = C.create_string_buffer(1024)
    rc = fn(*args, err, len(err))
    if rc != 0:
        raise DependencyError("ProcessError", err.value.decode() or f"error {rc}", code=-rc)


def predict_from_pandora_vcf(index_dir, pandora_vcf, out_vcf, opts=None):
    """Predict::predict_from_pandora_vcf: filter, minor-allele check, consequence, panel + expert-rule match."""
    opts = opts or AnnotateOpts.cli_defaults()
    _call(lib.drprg_hip_annotate, os.fsencode(index_dir), os.fsencode(pandora_vcf), os.fsencode(out_vcf), C.byref(opts))
    return out_vcf


def vcf_to_json(index_dir, vcf_path, json_path, sample="sample", padding=-1, index_version=None):
    """Predict::vcf_to_json"""
    _call(lib.drprg_hip_report_json, os.fsencode(index_dir), os.fsencode(vcf_path), os.fsencode(json_path), sample.encode(),
          padding, index_version.encode() if index_version is not None else None)
    return json_path
