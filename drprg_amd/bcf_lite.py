"""Tiny BCF2 reader (python, gzip + struct) used only to build synthetic test indexes from a panel.bcf;
the product reads panel.bcf in C++ (csrc/vcfio.cpp)."""
import gzip
import struct


def read_bcf(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"BCF\x02\x02" or raw[:4] == b"BCF\x02", "not a BCF2 file"
    l_text = struct.unpack_from("<I", raw, 5)[0]
    text = raw[9:9 + l_text].decode(errors="replace")
    contigs = [ln.split("ID=")[1].split(",")[0].split(">")[0] for ln in text.splitlines() if ln.startswith("##contig=")]
    p = 9 + l_text

    def typed(q):
        b = raw[q]
        q += 1
        t, n = b & 0xF, b >> 4
        if n == 15:
            tt, nn, q = typed(q)
            n = int.from_bytes(raw[q:q + {1: 1, 2: 2, 3: 4}[tt]], "little", signed=True)
            q += {1: 1, 2: 2, 3: 4}[tt]
        return t, n, q

    def string(q):
        t, n, q = typed(q)
        return raw[q:q + n].rstrip(b"\0").decode(), q + n

    recs = []
    while p + 8 <= len(raw):
        l_shared, l_indiv = struct.unpack_from("<II", raw, p)
        q = p + 8
        p = q + l_shared + l_indiv
        chrom, pos, rlen = struct.unpack_from("<iii", raw, q)
        n_allele_info = struct.unpack_from("<I", raw, q + 16)[0]
        q += 24
        vid, q = string(q)
        alleles = []
        for _ in range(n_allele_info >> 16):
            a, q = string(q)
            alleles.append(a)
        recs.append(dict(chrom=contigs[chrom], pos=pos, id=vid, ref=alleles[0], alts=alleles[1:]))
    return recs
