"""Independent BCF2 decoder (gzip + struct), test infrastructure only: decodes every field of every record back to VCF-like
values.  It is pinned on a BCF that htslib wrote -- the reference's own tests/cases/predict/panel.bcf (committed under
tests/golden/downstream/) -- and then used to check the product's BCF writer (csrc/bcfout.cpp)."""
import gzip
import re
import struct

INT_MISSING = {1: -128, 2: -32768, 3: -2147483648}
INT_EOV = {1: -127, 2: -32767, 3: -2147483647}
FLOAT_MISSING, FLOAT_EOV = 0x7F800001, 0x7F800002


def decode(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"BCF\x02\x02", "not BCF2.2"
    l_text = struct.unpack_from("<I", raw, 5)[0]
    text = raw[9:9 + l_text]
    assert text.endswith(b"\0")
    header = text.rstrip(b"\0").decode().splitlines()
    contigs, strings, kinds, nxt = [], {0: "PASS"}, {}, 1
    seen = {"PASS"}
    for ln in header:
        m = re.match(r"##(contig|FILTER|INFO|FORMAT)=<(.*)>", ln)
        if not m:
            continue
        attrs = dict(re.findall(r"(\w+)=(\"[^\"]*\"|[^,>]*)", m.group(2)))
        if m.group(1) == "contig":
            contigs.append(attrs["ID"])
            continue
        name = attrs["ID"]
        if "IDX" in attrs:
            strings[int(attrs["IDX"])] = name
            seen.add(name)
        elif name not in seen:
            seen.add(name)
            strings[nxt] = name
            nxt += 1
        kinds[(m.group(1), name)] = attrs.get("Type")
    p = 9 + l_text

    def typed(q):
        b = raw[q]
        q += 1
        t, n = b & 0xF, b >> 4
        if n == 15:
            tt, nn, q = typed(q)
            assert nn == 1
            n, q = ints(q, tt, 1)
            n = n[0]
        return t, n, q

    def ints(q, t, n):
        size, fmt = {1: (1, "b"), 2: (2, "h"), 3: (4, "i")}[t]
        vals = list(struct.unpack_from("<" + fmt * n, raw, q))
        return vals, q + size * n

    def value(q):
        t, n, q = typed(q)
        if t == 0 or n == 0:
            return None, q
        if t in (1, 2, 3):
            vals, q = ints(q, t, n)
            return [None if v == INT_MISSING[t] else v for v in vals if v != INT_EOV[t]], q
        if t == 5:
            bits = struct.unpack_from("<" + "I" * n, raw, q)
            vals = [None if b == FLOAT_MISSING else struct.unpack("<f", struct.pack("<I", b))[0] for b in bits if b != FLOAT_EOV]
            return vals, q + 4 * n
        if t == 7:
            return raw[q:q + n].rstrip(b"\0").decode(), q + n
        raise AssertionError(f"BCF type {t}")

    records = []
    while p + 8 <= len(raw):
        l_shared, l_indiv = struct.unpack_from("<II", raw, p)
        q = p + 8
        end_shared = q + l_shared
        end = end_shared + l_indiv
        chrom, pos, rlen = struct.unpack_from("<iii", raw, q)
        qual_bits = struct.unpack_from("<I", raw, q + 12)[0]
        n_allele_info, n_fmt_sample = struct.unpack_from("<II", raw, q + 16)
        q += 24
        rec = dict(chrom=contigs[chrom], pos=pos, rlen=rlen, qual=None if qual_bits == FLOAT_MISSING else struct.unpack("<f", struct.pack("<I", qual_bits))[0])
        vid, q = value(q)
        rec["id"] = vid or "."
        alleles = []
        for _ in range(n_allele_info >> 16):
            a, q = value(q)
            alleles.append(a or "")
        rec["alleles"] = alleles
        flt, q = value(q)
        rec["filters"] = [strings[i] for i in (flt or [])]
        info = []
        for _ in range(n_allele_info & 0xFFFF):
            key, q = value(q)
            v, q = value(q)
            info.append((strings[key[0]], v))
        rec["info"] = info
        assert q == end_shared, (q, end_shared)
        n_sample = n_fmt_sample & 0xFFFFFF
        fmt = []
        for _ in range(n_fmt_sample >> 24):
            key, q = value(q)
            name = strings[key[0]]
            t, n, q = typed(q)
            per_sample = []
            for _s in range(n_sample):
                if t in (1, 2, 3):
                    vals, q = ints(q, t, n)
                    if name == "GT":
                        per_sample.append("/".join("." if v >> 1 == 0 else str((v >> 1) - 1) for v in vals if v != INT_EOV[t]))
                    else:
                        per_sample.append([None if v == INT_MISSING[t] else v for v in vals if v != INT_EOV[t]])
                elif t == 5:
                    bits = struct.unpack_from("<" + "I" * n, raw, q)
                    q += 4 * n
                    per_sample.append([None if b == FLOAT_MISSING else struct.unpack("<f", struct.pack("<I", b))[0] for b in bits if b != FLOAT_EOV])
                elif t == 7:
                    per_sample.append(raw[q:q + n].rstrip(b"\0").decode())
                    q += n
                else:
                    raise AssertionError(f"BCF type {t}")
            fmt.append((name, per_sample))
        rec["format"] = fmt
        assert q == end, (q, end)
        records.append(rec)
        p = end
    return header, records
