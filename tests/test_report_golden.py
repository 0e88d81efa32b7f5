"""Annotate + report stage against the reference's own golden files (tests/golden/downstream/, copied from
/root/reference/tests/cases/predict/ by tools/make_golden.py).  Each case mirrors one reference test:
same options, same comparison (POS, sorted VARID, sorted PREDICT, and GT where the reference checks it; JSON compared
as whitespace-stripped strings)."""
import os

import pytest

from util import GOLDEN

DS = os.path.join(GOLDEN, "downstream")


def _opts(min_gt_conf=5.0, maf=0.1, max_gaps=0.3, max_gaps_diff=0.0, minor_min_covg=0, max_called_gaps=0.0,
          minor_min_strand_bias=0.0):
    from drprg_amd._lib import AnnotateOpts
    # Filterer { min_frs: 0.51, min_covg: 3, min_strand_bias: 0.01, max_indel: Some(20), min_gt_conf, ..Default }
    return AnnotateOpts(3, 2 ** 31 - 1, 0.01, min_gt_conf, 0.51, 20, maf, max_gaps, max_called_gaps, max_gaps_diff,
                        minor_min_covg, minor_min_strand_bias, 1, 12345)


def _records(path):
    out = []
    for line in open(path):
        if line.startswith("#"):
            continue
        t = line.rstrip("\n").split("\t")
        info = dict(kv.split("=", 1) if "=" in kv else (kv, "") for kv in t[7].split(";"))
        gt = t[9].split(":")[0] if len(t) > 9 else "."
        out.append(dict(chrom=t[0], pos=int(t[1]), varid=sorted(info.get("VARID", "").split(",")) if "VARID" in info else None,
                        predict=sorted(info.get("PREDICT", "").split(",")) if "PREDICT" in info else None,
                        gt=int(gt) if gt.isdigit() else -1, filt=t[6], info=info))
    return out


# (reference test, input, expected, annotate options, compare GT)
PREDICT_CASES = [
    # src/predict.rs:1554-1648: MinorAllele { maf 0.25, max_gaps 0.5, max_gaps_diff 0.3, minor_min_covg 3, max_called_gaps 0.39, sb 0.01 }
    ("test_predict_from_pandora_vcf", "in.vcf", "out.vcf",
     dict(maf=0.25, max_gaps=0.5, max_gaps_diff=0.3, minor_min_covg=3, max_called_gaps=0.39, minor_min_strand_bias=0.01), False),
    # :1653-1744, :1751-1854, :1862-1965: MinorAllele { maf 0.1, max_gaps 0.3, ..Default (zeros) }
    ("..._alt_that_is_susceptible_with_minor_resistance", "in2.vcf", "out2.vcf", dict(), False),
    ("..._alt_major_and_minor_with_unknowns", "in3.vcf", "out3.vcf", dict(), True),
    ("..._three_adjacent_mutations_only_one_called", "in4.vcf", "out4.vcf", dict(), True),
    # :1970-2075: min_gt_conf 0.0
    ("..._nullify_zero_depth_and_zero_confidence_calls", "ERR4796933.pandora.vcf", "ERR4796933.drprg.vcf", dict(min_gt_conf=0.0), True),
]


@pytest.mark.parametrize("name,inp,exp,kw,check_gt", PREDICT_CASES, ids=[c[1] for c in PREDICT_CASES])
def test_predict_from_pandora_vcf_goldens(tmp_path, name, inp, exp, kw, check_gt):
    from drprg_amd.predict import predict_from_pandora_vcf
    out = str(tmp_path / "test.drprg.vcf")
    predict_from_pandora_vcf(DS, os.path.join(DS, inp), out, _opts(**kw))
    actual, expected = _records(out), _records(os.path.join(DS, exp))
    assert len(actual) >= len(expected)
    for a, e in zip(actual, expected):
        where = f"{a['chrom']}:{a['pos']} vs {e['chrom']}:{e['pos']}"
        assert a["pos"] == e["pos"], where
        assert a["varid"] == e["varid"], where
        assert a["predict"] == e["predict"], where
        if check_gt:
            assert a["gt"] == e["gt"], where


def test_annotated_vcf_surface(tmp_path):
    """FILTER / INFO headers and PDP formatting equal the reference's out3.vcf; record IDs are 8 hex digits"""
    from drprg_amd.predict import predict_from_pandora_vcf
    out = str(tmp_path / "o.vcf")
    predict_from_pandora_vcf(DS, os.path.join(DS, "in3.vcf"), out, _opts())
    want = [l.rstrip("\n") for l in open(os.path.join(DS, "out3.vcf")) if l.startswith("##") and ("ID=ld" in l or "ID=sb" in l
            or "ID=lgc" in l or "ID=frs" in l or "ID=VARID" in l or "ID=PREDICT" in l or "ID=OGT" in l or "ID=PDP" in l)]
    got = [l.rstrip("\n") for l in open(out) if l.startswith("##")]
    for l in want:
        assert l in got, l
    assert any("ID=lindel,Description=\"Indel is longer than 20bp\"" in l for l in got)
    a, e = _records(out), _records(os.path.join(DS, "out3.vcf"))
    for x, y in zip(a, e):
        assert x["info"].get("PDP") == y["info"].get("PDP")
        assert x["filt"] == y["filt"]
    ids = [l.split("\t")[2] for l in open(out) if not l.startswith("#")]
    assert all(len(i) == 8 and int(i, 16) >= 0 for i in ids) and len(set(ids)) == len(ids)


JSON_CASES = [("out.vcf", "expected.json"), ("out3.vcf", "expected3.json"), ("out5.vcf", "expected5.json"),
              ("SRR6824468.vcf", "SRR6824468.json"), ("ERR4796933.drprg.vcf", "ERR4796933.json"),
              ("ERR2510634.drprg.vcf", "ERR2510634.json")]


@pytest.mark.parametrize("vcf,exp", JSON_CASES, ids=[c[0] for c in JSON_CASES])
def test_vcf_to_json_goldens(tmp_path, vcf, exp):
    """src/predict.rs:2078-2375: pred.vcf_to_json(vcf, 100, "version"), sample "test"; whitespace-stripped equality"""
    from drprg_amd.predict import vcf_to_json
    out = str(tmp_path / "test.drprg.json")
    vcf_to_json(DS, os.path.join(DS, vcf), out, sample="test", padding=100, index_version="version")
    strip = lambda s: "".join(s.split())
    assert strip(open(out).read()) == strip(open(os.path.join(DS, exp)).read())


def test_json_is_pretty_printed_like_serde(tmp_path):
    from drprg_amd.predict import vcf_to_json
    out = str(tmp_path / "j.json")
    vcf_to_json(DS, os.path.join(DS, "out3.vcf"), out, sample="test", padding=100, index_version="version")
    assert open(out).read() == open(os.path.join(DS, "expected3.json")).read().rstrip("\n")


def test_panel_bcf_reader_via_unknown_contig(tmp_path):
    """records on a contig the panel does not know are dropped (unwrap_or_continue!(name2rid), src/predict.rs:449)"""
    from drprg_amd.predict import predict_from_pandora_vcf
    src = open(os.path.join(DS, "in3.vcf")).read().splitlines()
    extra = [l for l in src if not l.startswith("#")][0].split("\t")
    extra[0] = "notagene"
    p = tmp_path / "in.vcf"
    p.write_text("\n".join(src + ["\t".join(extra)]) + "\n")
    out = str(tmp_path / "o.vcf")
    predict_from_pandora_vcf(DS, str(p), out, _opts())
    assert "notagene" not in {r["chrom"] for r in _records(out)}
