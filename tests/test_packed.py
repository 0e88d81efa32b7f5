"""2-bit packed reads, host side (include/drprg_hip.h "packed reads"; SURVEY.md section 8f NEXT-4 "optional 2-bit packing"; the reference takes
any fasta / fastq: /root/reference/src/predict.rs:166-170).  The device side is in tests/test_gpu_parity.py: every parity case runs
through both formats."""
import ctypes as C
import os

import numpy as np
import pytest

from util import ROOT


def _reference_pack(bases):
    """numpy restatement of the format: letter = bits 2:1 of the byte, 16 bases per word, first base lowest; positions of non-ACGTacgt bytes"""
    b = np.asarray(bases, np.uint8)
    letters = ((b >> 1) & 3).astype(np.uint64)
    pad = (-len(b)) % 16
    letters = np.concatenate([letters, np.zeros(pad, np.uint64)]).reshape(-1, 16)
    words = (letters << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
    up = b & 0xDF
    bad = ~((up == ord("A")) | (up == ord("C")) | (up == ord("G")) | (up == ord("T")))
    return words, np.nonzero(bad)[0].astype(np.uint64)


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 150, 1000, 4097, 100003])
def test_pack_reads_equals_the_format_definition(n):
    from drprg_amd.pandora import pack_reads
    rng = np.random.default_rng(n)
    b = np.frombuffer(b"ACGTacgtNnRY-*", np.uint8)[rng.choice(14, size=n, p=[0.2, 0.2, 0.2, 0.2, 0.04, 0.04, 0.04, 0.04, 0.01, 0.005, 0.005, 0.005, 0.005, 0.01])]
    words, npos = pack_reads(b)
    want_w, want_n = _reference_pack(b)
    assert np.array_equal(words, want_w) and np.array_equal(npos, want_n)


def test_pack_append_at_every_bit_offset():
    """the parser threads append read after read at arbitrary base offsets (pack.cpp pack_append): every (offset mod 32, length) pair
    against packing the concatenation at once -- through the packing ingest below; here the one-call helper on growing prefixes"""
    from drprg_amd.pandora import pack_reads
    rng = np.random.default_rng(3)
    b = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, size=700, p=[0.24, 0.24, 0.24, 0.24, 0.04])]
    for n in range(0, 700, 7):
        w, p = pack_reads(b[:n])
        ww, pp = _reference_pack(b[:n])
        assert np.array_equal(w, ww) and np.array_equal(p, pp), n


def _parse(path, threads, fmt):
    from drprg_amd._lib import lib
    out = (C.c_uint64 * 5)()
    err = C.create_string_buffer(512)
    old = os.environ.get("DRPRG_PARSE_FORMAT")
    os.environ["DRPRG_PARSE_FORMAT"] = fmt
    try:
        rc = lib.drprg_hip_parse_fastx(os.fsencode(path), threads, out, err, len(err))
    finally:
        if old is None:
            del os.environ["DRPRG_PARSE_FORMAT"]
        else:
            os.environ["DRPRG_PARSE_FORMAT"] = old
    assert rc == 0, err.value
    return tuple(int(x) for x in out[:3])


@pytest.mark.parametrize("kind", ["fastq", "fasta_multiline", "fastq_gz"])
def test_packing_ingest_keeps_every_read(tmp_path, kind):
    """drprg_hip_map_fastx with drprg_hip_set_input_format(ctx, 1): the parser threads pack the bases as they copy them.  Host-only
    self-check: reads / bases / order-independent digest of what the PACKED blocks say (letters back to bases, N at the recorded
    positions) equal those of the ASCII blocks with bases upper-cased and non-ACGT read as N -- ragged lengths (every bit offset),
    lower case, N and IUPAC codes, 1-16 threads, small blocks (many hand-overs)."""
    import gzip
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNRn", np.uint8)
    reads = []
    for i in range(30000):
        n = int(rng.integers(1, 400)) if i % 50 else int(rng.integers(0, 3))
        reads.append(alphabet[rng.choice(11, size=n, p=[0.22, 0.22, 0.22, 0.22, 0.02, 0.02, 0.02, 0.02, 0.02, 0.01, 0.01])].tobytes())
    path = str(tmp_path / ("r.fa" if kind == "fasta_multiline" else "r.fq"))
    with open(path, "wb") as fh:
        for i, r in enumerate(reads):
            if kind == "fasta_multiline":
                fh.write(b">r%d\n" % i)
                for j in range(0, len(r), 60):
                    fh.write(r[j:j + 60] + b"\n")
            else:
                fh.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    if kind == "fastq_gz":
        gz = path + ".gz"
        with gzip.open(gz, "wb", compresslevel=1) as fh:
            fh.write(open(path, "rb").read())
        path = gz
    os.environ["DRPRG_INGEST_BLOCK_MB"] = "1"
    try:
        want = _parse(path, 1, "normalized")
        assert want[0] == len(reads) and want[1] == sum(len(r) for r in reads)
        for threads in (1, 3, 8, 16):
            assert _parse(path, threads, "packed") == want, threads
    finally:
        del os.environ["DRPRG_INGEST_BLOCK_MB"]
