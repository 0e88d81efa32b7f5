"""Multi-threaded FASTA/FASTQ ingest (csrc/ingest.cpp) on CPU: every format variant parses to the same read multiset
whatever the thread count (digest = order-independent sum of per-read FNV-1a hashes)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse(path, threads):
    from drprg_amd._lib import lib
    out = (C.c_uint64 * 5)()
    err = C.create_string_buffer(512)
    rc = lib.drprg_hip_parse_fastx(os.fsencode(path), threads, out, err, len(err))
    if rc != 0:
        raise RuntimeError(f"{rc}: {err.value.decode()}")
    _parse.gz_mode = int(out[4])
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


# ---- gzip input: BGZF members in parallel (libdeflate), one plain member in one call, zlib streaming as the fallback ------
def _write_bgzf(path, data, block=65280, level=1):
    """bgzip's container: gzip members of <= 64 KB with a 'BC' extra field holding the member size - 1, then the empty EOF member"""
    import struct
    import zlib
    with open(path, "wb") as fh:
        for lo in list(range(0, len(data), block)) + [None]:
            chunk = data[lo:lo + block] if lo is not None else b""
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            bsize = 12 + 6 + len(body) + 8
            fh.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            fh.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def _random_fastq(n, seed, max_len=300):
    rng = np.random.default_rng(seed)
    reads, parts = [], []
    for i in range(n):
        r = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(rng.integers(1, max_len)), p=[.24, .26, .26, .23, .01]))
        reads.append(r)
        parts.append(b"@r%d some text\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    return reads, b"".join(parts)


@pytest.mark.parametrize("threads", [1, 4])
def test_gzip_inputs_take_the_fast_paths_and_agree(tmp_path, threads):
    import gzip
    reads, text = _random_fastq(120_000, 5)  # ~36 MB of text: more than one 32 MB window
    want = (len(reads), sum(len(r) for r in reads), _digest(reads))
    plain = tmp_path / "r.fq"
    plain.write_bytes(text)
    assert _parse(str(plain), threads)[:3] == want and _parse.gz_mode == 0
    bg = tmp_path / "r.bgzf.fq.gz"
    _write_bgzf(str(bg), text)
    assert _parse(str(bg), threads)[:3] == want and _parse.gz_mode == 1
    gz = tmp_path / "r.fq.gz"
    with gzip.open(gz, "wb", compresslevel=1) as fh:
        fh.write(text)
    # a plain gzip file of some size: with threads to spend all of them inflate the one stream (way 4), else one libdeflate call
    assert os.path.getsize(gz) > (4 << 20)
    assert _parse(str(gz), threads)[:3] == want and _parse.gz_mode == (4 if threads > 1 else 2)
    multi = tmp_path / "multi.fq.gz"  # several plain members (cat a.gz b.gz): way 4 follows them, and so does the streaming reader
    with open(multi, "wb") as fh:
        cut = text.index(b"\n@r60000 ") + 1
        fh.write(gzip.compress(text[:cut], 1) + gzip.compress(text[cut:], 1))
    assert _parse(str(multi), threads)[:3] == want and _parse.gz_mode == (4 if threads > 1 else 3)


def test_parallel_gunzip_can_be_switched_off_and_forced(tmp_path, monkeypatch):
    import gzip
    reads, text = _random_fastq(60_000, 6)
    want = (len(reads), sum(len(r) for r in reads), _digest(reads))
    gz = tmp_path / "r.fq.gz"
    gz.write_bytes(gzip.compress(text, 6))
    small = tmp_path / "small.fq.gz"
    small.write_bytes(gzip.compress(text[:text.index(b"\n@r2000 ") + 1], 6))
    assert _parse(str(small), 4)[0] == 2000 and _parse.gz_mode == 2      # too small to be worth the threads
    monkeypatch.setenv("DRPRG_GZ_PARALLEL", "1")
    monkeypatch.setenv("DRPRG_GZ_CHUNK", "30000")
    assert _parse(str(small), 4)[0] == 2000 and _parse.gz_mode == 4
    assert _parse(str(gz), 3)[:3] == want and _parse.gz_mode == 4
    monkeypatch.setenv("DRPRG_GZ_PARALLEL", "0")
    assert _parse(str(gz), 4)[:3] == want and _parse.gz_mode == 2
    # a damaged stream is refused on the parallel path too
    monkeypatch.setenv("DRPRG_GZ_PARALLEL", "1")
    raw = bytearray(gz.read_bytes())
    raw[len(raw) // 2] ^= 0x10
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(raw))
    with pytest.raises(RuntimeError):
        _parse(str(bad), 4)


def test_gzip_without_libdeflate_and_corrupt_blocks(tmp_path):
    import subprocess
    import sys
    reads, text = _random_fastq(3000, 9)
    bg = tmp_path / "r.fq.gz"
    _write_bgzf(str(bg), text)
    want = (len(reads), sum(len(r) for r in reads), _digest(reads))
    # a host without libdeflate: the same file through zlib (separate process: the library is bound once per process)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_ingest as t; "
            "r = t._parse(%r, 2); print(r[0], r[1], r[2], t._parse.gz_mode)" % (ROOT, os.path.join(ROOT, "tests"), str(bg)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, DRPRG_HIP_NO_LIBDEFLATE="1"))
    assert out.returncode == 0, out.stderr
    assert tuple(int(x) for x in out.stdout.split()) == want + (3,)
    # a flipped byte inside a member's deflate data: refused (CRC), not silently mis-parsed
    raw = bytearray(bg.read_bytes())
    raw[len(raw) // 2] ^= 0x55
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(raw))
    with pytest.raises(RuntimeError):
        _parse(str(bad), 2)
