"""<sample>.drprg.bcf: the product's BCF2.2 writer (csrc/bcfout.cpp; the reference writes BCF through rust-htslib,
/root/reference/src/predict.rs:429-431) checked by an independent decoder that is itself pinned on an htslib-written file."""
import ctypes as C
import os

import numpy as np
import pytest

from bcf_decode import decode
from util import GOLDEN

DOWN = os.path.join(GOLDEN, "downstream")


def test_decoder_reads_the_reference_panel_bcf():
    """tests/golden/downstream/panel.bcf = /root/reference/tests/cases/predict/panel.bcf (written by htslib): 1055 records over
    18 contigs, ID = <gene>_<variant>, INFO GENE/VAR/RES/DRUGS/PAD/ST; the product's C++ reader of the same
    file (drprg_amd.bcf_lite mirrors it) agrees on every record"""
    from drprg_amd.bcf_lite import read_bcf
    header, recs = decode(os.path.join(DOWN, "panel.bcf"))
    assert len(recs) == 1055 and len({r["chrom"] for r in recs}) == 18
    lite = read_bcf(os.path.join(DOWN, "panel.bcf"))
    assert [(r["chrom"], r["pos"], r["id"], r["alleles"][0], r["alleles"][1:]) for r in recs] == \
           [(r["chrom"], r["pos"], r["id"], r["ref"], r["alts"]) for r in lite]
    for r in recs[:200]:
        info = dict(r["info"])
        assert set(info) >= {"GENE", "VAR", "RES", "DRUGS"} and r["id"] == f"{info['GENE']}_{info['VAR']}"
        assert info["RES"] in ("DNA", "PROT") and r["rlen"] == len(r["alleles"][0])


def _to_bcf(vcf, bcf):
    from drprg_amd._lib import lib
    err = C.create_string_buffer(512)
    rc = lib.drprg_hip_vcf_to_bcf(os.fsencode(vcf), os.fsencode(bcf), err, len(err))
    assert rc == 0, err.value.decode()


def _vcf_records(path):
    header, recs = [], []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("##"):
            header.append(line)
        elif not line.startswith("#") and line:
            recs.append(line.split("\t"))
    return header, recs


def _num(x):
    return None if x in (".", "") else float(x)


@pytest.mark.parametrize("name", ["out.vcf", "out2.vcf", "out3.vcf", "out4.vcf", "ERR4796933.drprg.vcf", "SRR6824468.vcf", "ERR2510634.drprg.vcf"])
def test_bcf_writer_round_trips_the_reference_annotated_vcfs(tmp_path, name):
    """every annotated VCF among the reference's fixtures -> BCF -> decoded again: CHROM, POS, ID, alleles, FILTER, every INFO
    and FORMAT value equal the text (floats to f32 precision, which is what BCF stores)"""
    src = os.path.join(DOWN, name)
    bcf = str(tmp_path / (name + ".bcf"))
    _to_bcf(src, bcf)
    header, recs = decode(bcf)
    vh, vrecs = _vcf_records(src)
    assert header[-1].startswith("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t")
    assert [h for h in header[:-1] if not h.startswith("##FILTER=<ID=PASS")] == [h for h in vh if not h.startswith("##FILTER=<ID=PASS")]
    assert len(recs) == len(vrecs) > 0
    for r, t in zip(recs, vrecs):
        assert (r["chrom"], r["pos"] + 1, r["id"]) == (t[0], int(t[1]), t[2])
        assert r["alleles"] == [t[3]] + t[4].split(",") and r["rlen"] == len(t[3])
        assert r["filters"] == ([] if t[6] == "." else t[6].split(";"))
        want_info = [] if t[7] == "." else [kv.split("=", 1) if "=" in kv else [kv, None] for kv in t[7].split(";") if kv]
        assert [k for k, _ in r["info"]] == [k for k, _ in want_info]
        for (k, got), (_, want) in zip(r["info"], want_info):
            if isinstance(got, list):
                assert len(got) == len(want.split(","))
                for g, w in zip(got, want.split(",")):
                    assert (g is None and _num(w) is None) or abs(g - _num(w)) <= 1e-6 * max(1.0, abs(_num(w))), (k, got, want)
            else:
                assert got == want, (k, got, want)
        keys, vals = t[8].split(":"), t[9].split(":")
        assert [k for k, _ in r["format"]] == keys
        for (k, got), want in zip(r["format"], vals):
            got = got[0]
            if k == "GT":
                assert got == want
            elif isinstance(got, list):
                ws = want.split(",")
                assert len(got) == len(ws), (k, got, want)
                for g, w in zip(got, ws):
                    assert (g is None and _num(w) is None) or abs(g - _num(w)) <= 1e-6 * max(1.0, abs(_num(w))), (k, got, want)
            else:
                assert got == want


def test_bcf_is_a_bgzf_container_with_eof_marker(tmp_path):
    """BGZF: every member carries the 'BC' field with its own size, members hold <= 64 KB, the file ends with the 28-byte empty
    member that htslib checks for"""
    import struct
    bcf = str(tmp_path / "o.bcf")
    _to_bcf(os.path.join(DOWN, "out.vcf"), bcf)
    raw = open(bcf, "rb").read()
    off, members = 0, 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 14] == b"BC"
        size = struct.unpack_from("<H", raw, off + 16)[0] + 1
        isize = struct.unpack_from("<I", raw, off + size - 4)[0]
        assert isize <= 65536
        off += size
        members += 1
    assert off == len(raw) and members >= 2
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


@pytest.mark.parametrize("allele", [0, 62, 63, 200, 16383, 16384])
def test_genotype_of_a_very_multi_allelic_site_takes_the_width_it_needs(tmp_path, allele):
    """GT is stored as (allele + 1) << 1: allele 63 no longer fits the int8 the writer used for every GT (the byte read back as -128,
    the reserved 'missing'); htslib widens to int16 / int32, and so does the writer now"""
    n_alt = max(allele, 1)
    alts = ",".join("A" + "C" * (i % 7) + "G" * (i // 7 % 11) + "T" * (i // 77) for i in range(n_alt))
    vcf = tmp_path / "m.vcf"
    vcf.write_text("##fileformat=VCFv4.3\n##contig=<ID=g,length=100>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                   "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Depth\">\n"
                   "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample\n"
                   f"g\t5\t.\tT\t{alts}\t.\tPASS\t.\tGT:DP\t{allele}:2000000000\n"
                   f"g\t9\t.\tT\tA\t.\tPASS\t.\tGT:DP\t.:.\n")
    bcf = str(tmp_path / "m.bcf")
    _to_bcf(str(vcf), bcf)
    _, recs = decode(bcf)
    assert dict(recs[0]["format"])["GT"][0] == str(allele) and dict(recs[0]["format"])["DP"][0] == [2000000000]
    assert dict(recs[1]["format"])["GT"][0] == "." and len(recs[0]["alleles"]) == n_alt + 1
    # an integer that does not fit 32 bits is an error, not a wrapped value
    bad = tmp_path / "bad.vcf"
    bad.write_text(vcf.read_text().replace("2000000000", "5000000000"))
    from drprg_amd._lib import lib
    err = C.create_string_buffer(512)
    assert lib.drprg_hip_vcf_to_bcf(os.fsencode(str(bad)), os.fsencode(str(tmp_path / "bad.bcf")), err, len(err)) != 0 and b"32 bits" in err.value
