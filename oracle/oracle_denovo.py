"""oracle_denovo.py -- TEST INFRASTRUCTURE (the checker, never the product: only tests/ may import anything under oracle/).

A second, separately written statement of the accurate-read (-I) pile-up that stands in for the assembly half of `pandora discover`
(reference call site /root/reference/src/lib.rs:513-578; consumer: MakePrg::update, /root/reference/src/predict.rs:260-279).

PARITY UNPINNED, and more than that: this does NOT restate pandora.  pandora assembles the reads of a candidate region with a de
Bruijn graph (GATB; source not under /root/reference); the product uses a simpler exact-anchor pile-up (drprg_amd/csrc/denovo.cpp,
DESIGN.md section 4 "Discover"), and this file states the same RULES again in plain Python so that the product's multi-threaded,
hash-table implementation can be held against something independent: str.find over whole reads, no k-mer tables, no shared code.
Rules: the `anchor` consensus bases before and after a padded region are searched, exactly, in every read (>= 2 * anchor bases long)
and, reverse-complemented and swapped, for reads that run against the consensus; the first pair (left before right, at least an
anchor apart, spelled length within `max_len_change` of the region's) gives one allele per (region, orientation) and read; alleles
with an N are dropped; the most frequent allele (ties: the lexicographically smaller) is a novel variant if it differs from the
consensus and has >= min_support reads and >= min_fraction of the spanning reads; what it shares with the consensus on the left,
then on the right, is trimmed.
"""

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(_COMP.get(c, "N") for c in reversed(s))


def _occurrences(text, pat):
    out, at = [], text.find(pat)
    while at >= 0:
        out.append(at)
        at = text.find(pat, at + 1)
    return out


def pile_up(consensus, regions, reads, anchor=15, min_support=3, min_fraction=0.5, max_len_change=30):
    """consensus: {locus: called sequence}; regions: [(locus, start, end)] (0-based half-open, padding included); reads: iterable of
    str.  Returns [(locus, pos0, ref, alt, support, spanning)] sorted by (locus, pos0)."""
    todo = []
    for locus, start, end in regions:
        cons = consensus[locus]
        if start < anchor or end + anchor > len(cons):
            continue  # no room for an anchor on one side: left alone
        todo.append((locus, start, end, cons[start - anchor:start], cons[end:end + anchor], {}))
    for read in reads:
        read = read.upper()
        if len(read) < 2 * anchor:
            continue
        for locus, start, end, left, right, votes in todo:
            want = end - start
            for reverse in (False, True):
                first, second = (revcomp(right), revcomp(left)) if reverse else (left, right)
                span = None
                for x in _occurrences(read, first):
                    for y in _occurrences(read, second):
                        if y < x + anchor:
                            continue
                        got = y - (x + anchor)
                        if got > want + max_len_change or got + max_len_change < want:
                            continue
                        span = (x + anchor, y)
                        break
                    if span:
                        break
                if not span:
                    continue
                allele = read[span[0]:span[1]]
                if reverse:
                    allele = revcomp(allele)
                if "N" in allele or any(c not in "ACGT" for c in allele):
                    continue
                votes[allele] = votes.get(allele, 0) + 1
    out = []
    for locus, start, end, left, right, votes in todo:
        if not votes:
            continue
        spanning = sum(votes.values())
        best_n = max(votes.values())
        best = min(a for a, n in votes.items() if n == best_n)
        ref = consensus[locus][start:end]
        if best_n < min_support or best_n < min_fraction * spanning or best == ref:
            continue
        pre = 0
        while pre < len(ref) and pre < len(best) and ref[pre] == best[pre]:
            pre += 1
        suf = 0
        while suf < len(ref) - pre and suf < len(best) - pre and ref[len(ref) - 1 - suf] == best[len(best) - 1 - suf]:
            suf += 1
        out.append((locus, start + pre, ref[pre:len(ref) - suf], best[pre:len(best) - suf], best_n, spanning))
    return sorted(out, key=lambda v: (v[0], v[1]))
