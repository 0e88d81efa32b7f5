"""Multi-threaded FASTA/FASTQ ingest (csrc/ingest.cpp) on CPU: every format variant parses to the same read multiset
whatever the thread count (digest = order-independent sum of per-read FNV-1a hashes)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest


def _parse(path, threads):
    from drprg_amd._lib import lib
    out = (C.c_uint64 * 4)()
    err = C.create_string_buffer(512)
    rc = lib.drprg_hip_parse_fastx(os.fsencode(path), threads, out, err, len(err))
    if rc != 0:
        raise RuntimeError(f"{rc}: {err.value.decode()}")
    return int(out[0]), int(out[1]), int(out[2]), int(out[3])


def _digest(reads):
    s = 0
    for r in reads:
        h = 1469598103934665603
        for b in r:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        s = (s + h) & 0xFFFFFFFFFFFFFFFF
    return s


@pytest.fixture(scope="module")
def reads():
    rng = np.random.default_rng(5)
    out = []
    for i in range(3000):
        L = int(rng.choice([0, 1, 30, 150, 151, 1000, 5000]))
        out.append(bytes(rng.choice(list(b"ACGTN"), size=L, p=[0.24, 0.25, 0.25, 0.24, 0.02]).tolist()))
    return out


def _fastq(reads, nl=b"\n", tricky_quals=True):
    recs = []
    for i, r in enumerate(reads):
        q = (b"@" if tricky_quals and i % 3 == 0 and r else b"I") + b"+" * max(0, len(r) - 1)  # quality lines starting with '@' / full of '+'
        recs.append(b"@r%d some comment" % i + nl + r + nl + b"+" + nl + q[:len(r)] + nl)
    return b"".join(recs)


def test_fastq_plain_gz_threads(tmp_path, reads):
    want = (len(reads), sum(len(r) for r in reads), _digest(reads))
    data = _fastq(reads)
    p = tmp_path / "r.fq"
    p.write_bytes(data)
    pz = tmp_path / "r.fq.gz"
    with gzip.open(pz, "wb", compresslevel=1) as fh:
        fh.write(data)
    for path in (p, pz):
        for t in (1, 2, 7):
            got = _parse(str(path), t)
            assert got[:3] == want, (path, t)


def test_fastq_crlf_and_no_trailing_newline(tmp_path, reads):
    want = (len(reads), sum(len(r) for r in reads), _digest(reads))
    p = tmp_path / "crlf.fq"
    p.write_bytes(_fastq(reads, nl=b"\r\n"))
    assert _parse(str(p), 3)[:3] == want
    p2 = tmp_path / "nonl.fq"
    p2.write_bytes(_fastq(reads).rstrip(b"\n"))
    assert _parse(str(p2), 3)[:3] == want


def test_fasta_multiline(tmp_path, reads):
    want = (len(reads), sum(len(r) for r in reads), _digest(reads))
    recs = []
    for i, r in enumerate(reads):
        recs.append(b">s%d\n" % i + b"\n".join(r[j:j + 60] for j in range(0, len(r), 60)) + b"\n")
    p = tmp_path / "r.fa"
    p.write_bytes(b"".join(recs))
    for t in (1, 4):
        assert _parse(str(p), t)[:3] == want


def test_multiline_fastq_falls_back_to_the_serial_reader(tmp_path, reads):
    sub = [r for r in reads if len(r) >= 30][:200]
    want = (len(sub), sum(len(r) for r in sub), _digest(sub))
    recs = []
    for i, r in enumerate(sub):
        h = len(r) // 2
        recs.append(b"@m%d\n" % i + r[:h] + b"\n" + r[h:] + b"\n+\n" + b"I" * h + b"\n" + b"I" * (len(r) - h) + b"\n")
    p = tmp_path / "ml.fq"
    p.write_bytes(b"".join(recs))
    assert _parse(str(p), 4)[:3] == want


def test_large_file_is_cut_into_slices(tmp_path):
    """> 96 MB of text: several slices, several batches, identical digest for 1 and 8 threads"""
    rng = np.random.default_rng(1)
    n, L = 400_000, 150
    bases = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n * L)
    from drprg_amd import synth
    p = str(tmp_path / "big.fq")
    synth.write_fastq_fixed(p, bases, L)
    a, b = _parse(p, 1), _parse(p, 8)
    assert a[:3] == b[:3] and a[0] == n and a[1] == n * L
    assert os.path.getsize(p) > 120e6


def test_malformed_inputs(tmp_path):
    p = tmp_path / "bad.fq"
    p.write_bytes(b"@r1\nACGT\n+\nIIII\n@r2\nACGT\nIIII\n@r3\nAC\n+\nII\n")
    with pytest.raises(RuntimeError):
        _parse(str(p), 2)
    p2 = tmp_path / "notfastx.txt"
    p2.write_bytes(b"hello world\n")
    with pytest.raises(RuntimeError):
        _parse(str(p2), 2)
    with pytest.raises(RuntimeError):
        _parse(str(tmp_path / "missing.fq"), 2)
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    assert _parse(str(empty), 2)[:2] == (0, 0)
