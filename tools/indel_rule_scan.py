#!/usr/bin/env python3
"""Which allele -> k-mer rule do the reference's fixtures admit for the records pandora prints with a padding base (indels)?

tests/golden/kmer_count_kat.tsv holds, per allele of the seven pandora VCFs of /root/reference/tests/cases/predict/, the k-mer counts its
MEAN / SUM / GAPS (and, since round 5, MED) fields admit.  The default rule (k-mers at bases s of the allele's route with s < B and
s + k >= A, [A, B) = the allele as printed) explains 318 of 324 informative alleles; two of the six misses are single-base deletions
(in.vcf embA:69 ALT, gid:160 REF).  This script scores a family of rules for padded records over ALL informative
alleles -- the default rule stays for records without a padding base -- and prints the table DESIGN.md section 5 quotes.

usage: python tools/indel_rule_scan.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kmer_count_kat as T  # noqa: E402
from util import Oracle  # noqa: E402


def is_padded(r):
    alls = [r["ref"]] + r["alts"]
    return len({len(a) for a in alls}) > 1 and all(a[:1] == r["ref"][:1] for a in alls) and min(len(a) for a in alls) == 1


def score(orc):
    tot = ok = ind_tot = ind_ok = 0
    miss = []
    for f, w in T.FILE_W.items():
        padded = {(r["chrom"], r["pos"]) for r in T.KAT[f] if is_padded(r)}
        for (chrom, pos, a), (n, feas) in T._oracle_counts(orc, f, w).items():
            hit = n in feas
            tot += 1
            ok += hit
            if (chrom, pos) in padded:
                ind_tot += 1
                ind_ok += hit
            if not hit:
                miss.append(f"{f.split('.')[0]}:{chrom}:{pos}:{a}({n} not in {feas[:3]})")
    return ok, tot, ind_ok, ind_tot, miss


def main():
    orc = Oracle()
    L = orc.lib
    rows = []
    try:
        L.orc_vcf_set_overlap_rule(0)
        rows.append(("default: printed range [A, B), s < B and s + k >= A", score(orc)))
        L.orc_vcf_set_overlap_rule(2)
        for bare in (0, 1):
            for dl in (-1, 0, 1, 2):
                for dr in (-1, 0, 1):
                    for ext in (0, 1):
                        L.orc_vcf_set_padded_rule(bare, dl, dr, ext)
                        name = f"padded records: {'bare' if bare else 'printed'} range, s + k >= R0 + {dl}, s < R1 + {dr}" + (", an empty allele one base wide" if ext else "")
                        rows.append((name, score(orc)))
        L.orc_vcf_set_overlap_rule(3)
        for dl in (-2, -1, 0, 1, 2):
            for dr in (-2, -1, 0, 1, 2):
                L.orc_vcf_set_all_rule(dl, dr)
                rows.append((f"EVERY record: printed range, s + k >= A + {dl}, s < B + {dr}", score(orc)))
    finally:
        L.orc_vcf_set_overlap_rule(0)
        L.orc_vcf_set_padded_rule(0, 0, 0, 0)
        L.orc_vcf_set_all_rule(0, 0)
    rows.sort(key=lambda r: (-r[1][0], r[0]))
    print(f"{'rule for records with a padding base':100s} all alleles   padded-record alleles   misses")
    for name, (ok, tot, iok, itot, miss) in rows:
        print(f"{name:100s} {ok:3d} / {tot:3d}     {iok:2d} / {itot:2d}    {' '.join(miss) if len(miss) <= 9 else str(len(miss)) + ' misses'}")


if __name__ == "__main__":
    main()
