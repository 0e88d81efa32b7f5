"""One gzip stream inflated by several threads (csrc/pgunzip.cpp, way 4 of the ingest): every kind of deflate block, member
layout and chunk size gives the bytes zlib gives, and what gzip refuses is refused."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest


def _gunzip(path, out, threads=4, chunk=0):
    from drprg_amd._lib import lib
    o = (C.c_uint64 * 3)()
    err = C.create_string_buffer(512)
    rc = lib.drprg_hip_gunzip_file(os.fsencode(str(path)), threads, chunk, os.fsencode(str(out)), o, err, len(err))
    if rc != 0:
        raise RuntimeError(f"{rc}: {err.value.decode()}")
    return int(o[0]), int(o[1]), int(o[2])


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush_mode=zlib.Z_SYNC_FLUSH, name=None, extra=None,
            comment=None, hcrc=False):
    """a gzip member written by hand: any header field, any zlib strategy, optional flush points inside"""
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03"
    if extra is not None:
        head += struct.pack("<H", len(extra)) + extra
    if name is not None:
        head += name + b"\0"
    if comment is not None:
        head += comment + b"\0"
    if hcrc:
        head += struct.pack("<H", zlib.crc32(head) & 0xFFFF)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = b""
    if flush_every:
        for i in range(0, len(data), flush_every):
            body += co.compress(data[i:i + flush_every]) + co.flush(flush_mode)
    else:
        body = co.compress(data)
    body += co.flush()
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def _fastq_text(n, seed=1, read_len=150):
    rng = np.random.default_rng(seed)
    seqs = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n, read_len))]
    quals = (rng.integers(20, 41, size=(n, read_len)) + 33).astype(np.uint8)
    out = []
    for i in range(n):
        out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ATCACG\n" % (1101 + i // 5000, 1000 + i % 977, 2000 + i % 3571)
                   + seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    return b"".join(out)


@pytest.fixture(scope="module")
def text():
    return _fastq_text(40000)  # 14 MB


def _check(tmp_path, blob, want, threads=4, chunk=0, name="t.gz"):
    p = tmp_path / name
    p.write_bytes(blob)
    o = tmp_path / (name + ".out")
    n, accepted, redone = _gunzip(p, o, threads, chunk)
    got = o.read_bytes()
    assert n == len(want) and got == want, (n, len(want))
    assert gzip.decompress(blob) == want  # (the file itself is sound)
    return accepted, redone


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("chunk", [0, 4096, 70000, 1 << 20])
def test_fastq_text_every_level_and_chunk_size(tmp_path, text, level, chunk):
    accepted, redone = _check(tmp_path, gzip.compress(text, level), text, chunk=chunk)
    if chunk == 70000:  # blocks are shorter than the chunks: every chunk's thread finds its own way in, nothing is inflated twice
        assert accepted > 30 and redone == 0


@pytest.mark.parametrize("threads", [1, 2, 3, 8, 16])
def test_thread_counts(tmp_path, text, threads):
    _check(tmp_path, gzip.compress(text, 6), text, threads=threads, chunk=100000)


def test_stored_fixed_and_odd_strategies(tmp_path, text):
    rng = np.random.default_rng(2)
    noise = rng.integers(0, 256, size=3_000_000, dtype=np.uint8).tobytes()
    small = text[:2_000_000]
    cases = {
        "stored": _member(noise, level=0),                                   # stored blocks only
        "noise6": _member(noise, level=6),                                   # incompressible: zlib mixes stored and dynamic blocks
        "fixed": _member(small, strategy=zlib.Z_FIXED),                      # fixed-Huffman blocks only: no dynamic header to find
        "huffman": _member(small, strategy=zlib.Z_HUFFMAN_ONLY),             # no matches at all: empty distance code
        "rle": _member(small, strategy=zlib.Z_RLE),                          # distance 1 only: a single distance code
        "mixed": _member(small + noise + small, level=6),
    }
    want = {"stored": noise, "noise6": noise, "fixed": small, "huffman": small, "rle": small, "mixed": small + noise + small}
    for k, blob in cases.items():
        for chunk in (5000, 200000):
            _check(tmp_path, blob, want[k], chunk=chunk, name=k + ".gz")


def test_long_matches_and_long_blocks(tmp_path):
    zeros = bytes(60_000_000)                                # 1000:1 -- blocks far longer than a chunk of text, the soft cap of a segment
    _check(tmp_path, gzip.compress(zeros, 6), zeros, chunk=8192)
    rng = np.random.default_rng(3)
    unit = rng.integers(0, 256, size=32768, dtype=np.uint8).tobytes()
    rep = unit * 300                                         # every match at distance 32768: the whole window is in use
    _check(tmp_path, gzip.compress(rep, 9), rep, chunk=3000, name="rep.gz")


def test_members_headers_flushes_and_padding(tmp_path, text):
    a, b, c = text[:3_000_000], text[3_000_000:3_000_001], text[3_000_001:9_000_000]
    blob = (_member(a, name=b"reads.fq", comment=b"lane 1", extra=b"AB\x02\x00xy", hcrc=True) + _member(b"") + _member(b, level=9)
            + _member(c, flush_every=70000) + _member(b"") + _member(text[9_000_000:], flush_every=50000, flush_mode=zlib.Z_FULL_FLUSH))
    for chunk in (3000, 64000, 1 << 20):
        _check(tmp_path, blob, text, chunk=chunk, name="multi.gz")
    # zero padding and foreign bytes after the last member are ignored, as gzip does (with a warning)
    p = tmp_path / "pad.gz"
    p.write_bytes(blob + bytes(5000))
    assert _gunzip(p, tmp_path / "pad.out", 4, 50000)[0] == len(text) and (tmp_path / "pad.out").read_bytes() == text
    p.write_bytes(blob + b"trailing garbage that is not a member")
    assert _gunzip(p, tmp_path / "pad.out", 4, 50000)[0] == len(text)
    # a file of empty members only
    p.write_bytes(_member(b"") * 3)
    assert _gunzip(p, tmp_path / "pad.out", 4, 1024)[0] == 0


def test_damaged_streams_are_refused(tmp_path, text):
    blob = gzip.compress(text[:6_000_000], 6)
    p = tmp_path / "bad.gz"
    out = tmp_path / "bad.out"
    for cut in (len(blob) - 1, len(blob) - 8, len(blob) // 2, 20):  # truncated inside the trailer, at it, in the data, after the header
        p.write_bytes(blob[:cut])
        with pytest.raises(RuntimeError):
            _gunzip(p, out, 4, 60000)
    wrong_crc = bytearray(blob)
    wrong_crc[-6] ^= 1
    p.write_bytes(bytes(wrong_crc))
    with pytest.raises(RuntimeError, match="CRC"):
        _gunzip(p, out, 4, 60000)
    wrong_len = bytearray(blob)
    wrong_len[-1] ^= 1
    p.write_bytes(bytes(wrong_len))
    with pytest.raises(RuntimeError, match="CRC"):
        _gunzip(p, out, 4, 60000)
    rng = np.random.default_rng(4)
    for _ in range(12):  # a flipped bit anywhere in the deflate data: a decoding error or the CRC, never wrong bytes handed out as good
        hit = bytearray(blob)
        at = int(rng.integers(12, len(blob) - 8))
        hit[at] ^= 1 << int(rng.integers(0, 8))
        p.write_bytes(bytes(hit))
        try:
            zlib_ok = gzip.decompress(bytes(hit)) == text[:6_000_000]
        except Exception:
            zlib_ok = False
        if zlib_ok:
            continue  # (a flip in a stored block's padding bits changes nothing)
        with pytest.raises(RuntimeError):
            _gunzip(p, out, 4, 60000)
    p.write_bytes(b"\x1f\x8b\x08\x00" + bytes(30))
    with pytest.raises(RuntimeError):
        _gunzip(p, out, 2, 0)
    p.write_bytes(b"not gzip at all, just text\n" * 10)
    with pytest.raises(RuntimeError):
        _gunzip(p, out, 2, 0)


def test_header_look_alikes_do_not_mislead(tmp_path):
    """the compressed bytes of random data hold every bit pattern: chunk boundaries land among stored-block payload that the
    finder must search through; whatever it picks, the stitcher only keeps chunks that start where their predecessor ended"""
    rng = np.random.default_rng(9)
    parts = []
    for i in range(40):
        parts.append(rng.integers(0, 256, size=int(rng.integers(1000, 200000)), dtype=np.uint8).tobytes())
        parts.append(_fastq_text(int(rng.integers(10, 600)), seed=100 + i))
    data = b"".join(parts)
    blob = gzip.compress(data, 6)
    for chunk in (2048, 10000, 50000):
        _check(tmp_path, blob, data, chunk=chunk, name="alike.gz")


def test_bgzf_file_as_a_plain_multi_member_stream(tmp_path, text):
    """a bgzip file is hundreds of 64 KB members (each with an extra field) and an empty one at the end: the ingest has a
    faster way for those, but this decoder must take them too (a host without libdeflate sends them here)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    p = tmp_path / "t.bgzf.gz"
    bench.write_bgzf(str(p), text[:8_000_000])
    for chunk in (20000, 300000):
        n, accepted, redone = _gunzip(p, tmp_path / "o", 4, chunk)
        assert n == 8_000_000 and (tmp_path / "o").read_bytes() == text[:8_000_000]


@pytest.mark.parametrize("seed", range(6))
def test_random_member_layouts(tmp_path, seed):
    """random mixes of text, noise, runs and empty pieces, compressed member by member with random levels, strategies, flush
    points and header fields, then inflated with a random chunk size and thread count: always the bytes zlib gives"""
    rng = np.random.default_rng(1000 + seed)
    pieces, blob = [], b""
    for m in range(int(rng.integers(1, 7))):
        kind = int(rng.integers(0, 5))
        n = int(rng.integers(0, 1_500_000))
        if kind == 0:
            data = _fastq_text(n // 330 + 1, seed=seed * 50 + m)
        elif kind == 1:
            data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif kind == 2:
            data = bytes([int(rng.integers(0, 256))]) * n
        elif kind == 3:
            unit = rng.integers(65, 91, size=int(rng.integers(1, 40000)), dtype=np.uint8).tobytes()
            data = (unit * (n // len(unit) + 1))[:n]
        else:
            data = b""
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
        flush_every = [0, 0, 4096, 100_000][int(rng.integers(0, 4))]
        blob += _member(data, level=int(rng.integers(0, 10)), strategy=strategy, flush_every=flush_every,
                        flush_mode=[zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH][int(rng.integers(0, 2))],
                        name=b"x.fq" if rng.random() < 0.3 else None, hcrc=bool(rng.random() < 0.3))
        pieces.append(data)
    want = b"".join(pieces)
    chunk = [0, 1500, 20_000, 300_000][int(rng.integers(0, 4))]
    _check(tmp_path, blob, want, threads=int(rng.integers(1, 9)), chunk=chunk, name=f"rand{seed}.gz")


@pytest.mark.parametrize("cap", [300, 5000, 70000])
def test_a_small_symbol_cap_and_flush_points_everywhere(tmp_path, monkeypatch, cap):
    """chunks that stop at the symbol cap are taken up again from where they stopped; with flush points (empty stored blocks) every
    few hundred bytes some of those continuations hold no text at all -- a segment like that is skipped, it does not end the stream"""
    rng = np.random.default_rng(cap)
    monkeypatch.setenv("DRPRG_GZ_PARALLEL", "1")
    monkeypatch.setenv("DRPRG_GZ_SOFT_CAP", str(cap))
    pieces = []
    for i in range(60):
        kind = int(rng.integers(0, 3))
        n = int(rng.integers(1, 40000))
        pieces.append(_fastq_text(n // 330 + 1, seed=i)[:n] if kind == 0 else bytes(n) if kind == 1 else rng.integers(0, 256, size=n, dtype=np.uint8).tobytes())
    data = b"".join(pieces)
    blob = b"".join(_member(data[a:a + 400000], level=int(rng.integers(1, 10)), flush_every=int(rng.integers(150, 3000)),
                            flush_mode=[zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH][int(rng.integers(0, 2))]) for a in range(0, len(data), 400000))
    gz = tmp_path / "f.gz"
    gz.write_bytes(blob)
    for threads, chunk in ((3, 2048), (8, 4096), (5, 30000)):
        out = tmp_path / f"o{threads}"
        n, _, redone = _gunzip(gz, out, threads=threads, chunk=chunk)
        assert n == len(data) and out.read_bytes() == data
