"""Synthetic PRG panels, genomes and reads for parity tests and bench.py.

The real `mtb` index and real reads are not obtainable offline, so the workloads BASELINE.json names
are generated: an "mtb-like" panel (18 loci with the names and padded lengths of
/root/reference/tests/cases/predict/genes.fa.fai, k=15, w=11), a background genome of 4,411,532 bp at
65.6 % GC with the loci implanted, and reads sampled from haplotype copies of that genome.
Everything is seeded and deterministic.
"""
import numpy as np

# names and padded lengths of the 18 genes of the reference's test index (genes.fa.fai)
MTB_LOCI = [("embA", 3485), ("fabG1", 944), ("rpsL", 575), ("gid", 875), ("rplC", 854), ("eis", 1409), ("rrs", 1737),
            ("rpoB", 3719), ("ethA", 1670), ("ahpC", 788), ("ddn", 656), ("gyrB", 2228), ("tlyA", 1007), ("embB", 3497),
            ("gyrA", 2717), ("pncA", 761), ("katG", 2423), ("inhA", 1010)]
MTB_GENOME_SIZE = 4411532
import os as _os
MTB_8D_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "data", "mtb_8d")  # genes.fa + panel.bcf of the reference's test index
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGTNacgtn")] = list(b"TGCANtgcan")


def random_seq(rng, n, gc=0.656):
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    return _ACGT[rng.choice(4, size=n, p=p)].tobytes().decode()


class Site:
    """alleles: list of segment lists; a segment is a str or a nested Site"""

    def __init__(self, alleles):
        self.alleles = alleles


def _mutate(rng, s, indel):
    """an alternative allele for reference allele s"""
    if indel:
        if rng.random() < 0.5 and len(s) > 0:
            return s[: int(rng.integers(0, len(s)))]  # deletion of the tail (possibly the empty allele)
        return s + random_seq(rng, int(rng.integers(1, 21)))
    if not s:
        return random_seq(rng, 1)
    i = int(rng.integers(0, len(s)))
    alt = "ACGT".replace(s[i], "")[int(rng.integers(0, 3))]
    return s[:i] + alt + s[i + 1:]


def make_locus(rng, length, site_every=60, max_alts=4, nested_frac=0.1, indel_frac=0.05, gc=0.656):
    """Segment list for one locus: backbone of ~length bases with a site about every `site_every` bases."""
    segs = []
    pos = 0
    backbone = random_seq(rng, length, gc)
    while pos < length:
        gap = int(rng.integers(max(8, site_every // 2), site_every * 3 // 2 + 1))
        if pos == 0:
            gap = max(gap, 30)  # keep the first minimizer windows clear of variation
        nxt = min(length, pos + gap)
        segs.append(backbone[pos:nxt])
        pos = nxt
        if pos >= length - 30:
            segs[-1] += backbone[pos:]
            break
        ref_len = int(rng.integers(1, 5))
        ref = backbone[pos:pos + ref_len]
        pos += ref_len
        n_alts = int(rng.integers(1, max_alts + 1))
        alleles, seen = [[ref]], {ref}
        for _ in range(n_alts):
            for _try in range(8):
                alt = _mutate(rng, ref, rng.random() < indel_frac)
                if alt not in seen:
                    seen.add(alt)
                    alleles.append([alt])
                    break
        if rng.random() < nested_frac and len(ref) >= 1:
            # depth-2: an alternative long allele holding its own SNP site
            inner_ref = random_seq(rng, 1, gc)
            inner_alt = "ACGT".replace(inner_ref, "")[int(rng.integers(0, 3))]
            alleles.append([random_seq(rng, int(rng.integers(0, 6)), gc), Site([[inner_ref], [inner_alt]]),
                            random_seq(rng, int(rng.integers(2, 9)), gc)])
        if len(alleles) >= 2:
            segs.append(Site(alleles))
        else:
            segs[-1] += ref
    # merge adjacent strings
    out = []
    for s in segs:
        if isinstance(s, str) and out and isinstance(out[-1], str):
            out[-1] += s
        else:
            out.append(s)
    if isinstance(out[-1], Site):
        out.append("")
    return out


def prg_string(segs):
    """make_prg syntax: odd marker opens/closes a site, the next even one separates alleles; pre-order numbering from 5."""
    counter = [5]

    def emit(seglist):
        s = ""
        for seg in seglist:
            if isinstance(seg, str):
                s += seg
            else:
                m = counter[0]
                counter[0] += 2
                parts = [emit(a) for a in seg.alleles]
                s += f" {m} " + f" {m + 1} ".join(parts) + f" {m} "
        return s

    return emit(segs)


def sample_haplotype(rng, segs, first_allele=False):
    s = ""
    for seg in segs:
        if isinstance(seg, str):
            s += seg
        else:
            a = seg.alleles[0] if first_allele else seg.alleles[int(rng.integers(0, len(seg.alleles)))]
            s += sample_haplotype(rng, a, first_allele)
    return s


class Panel:
    """A set of loci: PRG strings, reference (first-allele) sequences, segment trees."""

    def __init__(self, names, trees):
        self.names = names
        self.trees = trees
        self.prgs = [prg_string(t) for t in trees]
        self.refs = [sample_haplotype(None, t, first_allele=True) for t in trees]

    def write(self, prg_path, genes_fa_path=None):
        with open(prg_path, "w") as fh:
            for n, p in zip(self.names, self.prgs):
                fh.write(f">{n}\n{p}\n")
        if genes_fa_path:
            with open(genes_fa_path, "w") as fh:
                for n, r in zip(self.names, self.refs):
                    fh.write(f">{n} padding=100\n{r}\n")


def mtb_like_panel(seed=20230308, site_every=60):
    rng = np.random.default_rng(seed)
    return Panel([n for n, _ in MTB_LOCI], [make_locus(rng, L, site_every=site_every) for _, L in MTB_LOCI])


def big_panel(seed=500, n_loci=500, n_sites=50000):
    """config 5: 500 loci of length ~U(600,3500), ~1 site per 20 bp, 2-5 alleles, 5 % indels, 10 % nested."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(600, 3501, size=n_loci)
    site_every = max(12, int(lens.sum() / n_sites))
    return Panel([f"locus{i:03d}" for i in range(n_loci)],
                 [make_locus(rng, int(L), site_every=site_every, max_alts=4, gc=0.65) for L in lens])


def small_panel(seed=7, n_loci=4, length=700, site_every=40):
    rng = np.random.default_rng(seed)
    return Panel([f"g{i}" for i in range(n_loci)], [make_locus(rng, length, site_every=site_every, nested_frac=0.2,
                                                               indel_frac=0.15) for _ in range(n_loci)])


class HaplotypeGenomes:
    """n_hap copies of a background genome with every locus replaced by a sampled haplotype."""

    def __init__(self, panel, genome_size=MTB_GENOME_SIZE, n_hap=16, seed=4411532, gc=0.656):
        rng = np.random.default_rng(seed)
        total = sum(len(r) for r in panel.refs)
        if genome_size < 2 * total + 1000:
            genome_size = 2 * total + 1000
        n = len(panel.refs)
        spacer = (genome_size - total) // (n + 1)
        bg = [random_seq(rng, spacer, gc) for _ in range(n + 1)]
        self.haps = []
        for _ in range(n_hap):
            parts = []
            for i in range(n):
                parts.append(bg[i])
                parts.append(sample_haplotype(rng, panel.trees[i]))
            parts.append(bg[n])
            self.haps.append(np.frombuffer("".join(parts).encode(), dtype=np.uint8))
        self.lens = np.array([len(h) for h in self.haps], dtype=np.int64)
        self.max_len = int(self.lens.max())

    def padded(self):
        a = np.full((len(self.haps), self.max_len), ord("N"), dtype=np.uint8)
        for i, h in enumerate(self.haps):
            a[i, :len(h)] = h
        return a


def sample_short_reads(genomes, n_reads, read_len=150, seed=1, sub_rate=0.001, chunk=1 << 20):
    """Uniform 150 bp reads, random strand, substitution errors.  Returns (bases u8[n*L], offsets u64[n+1])."""
    rng = np.random.default_rng(seed)
    out = np.empty(n_reads * read_len, dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    for lo in range(0, n_reads, chunk):
        m = min(chunk, n_reads - lo)
        hap = rng.integers(0, len(genomes.haps), size=m)
        start = (rng.random(m) * (genomes.lens[hap] - read_len)).astype(np.int64)
        rev = rng.random(m) < 0.5
        block = np.empty((m, read_len), dtype=np.uint8)
        for h in range(len(genomes.haps)):
            sel = np.nonzero(hap == h)[0]
            if sel.size:
                block[sel] = genomes.haps[h][start[sel, None] + ar]
        block[rev] = _COMP[block[rev][:, ::-1]]
        nerr = rng.binomial(m * read_len, sub_rate)
        if nerr:
            pos = rng.integers(0, m * read_len, size=nerr)
            flat = block.reshape(-1)
            flat[pos] = _ACGT[(np.searchsorted(_ACGT, flat[pos]) + rng.integers(1, 4, size=nerr)) % 4]
        out[lo * read_len:(lo + m) * read_len] = block.reshape(-1)
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return out, offsets


def sample_long_reads(genomes, n_reads, seed=3, mean_len=4000, sigma=0.5, min_len=500, max_len=50000, err=0.05):
    """Nanopore-like reads: lognormal lengths, 5 % errors split 40/30/30 substitution/insertion/deletion."""
    rng = np.random.default_rng(seed)
    mu = np.log(mean_len) - sigma * sigma / 2
    lens = np.clip(rng.lognormal(mu, sigma, size=n_reads), min_len, max_len).astype(np.int64)
    hap = rng.integers(0, len(genomes.haps), size=n_reads)
    lens = np.minimum(lens, genomes.lens[hap] - 1)
    start = (rng.random(n_reads) * (genomes.lens[hap] - lens)).astype(np.int64)
    rev = rng.random(n_reads) < 0.5
    pieces, offsets = [], [0]
    for i in range(n_reads):
        t = genomes.haps[hap[i]][start[i]:start[i] + lens[i]]
        if rev[i]:
            t = _COMP[t[::-1]]
        u = rng.random(t.size)
        counts = np.ones(t.size, dtype=np.int64)
        counts[u < err * 0.3] = 0                               # deletion
        ins = (u >= err * 0.3) & (u < err * 0.6)
        counts[ins] = 2                                         # insertion after this base
        sub = (u >= err * 0.6) & (u < err)
        t = t.copy()
        if sub.any():
            t[sub] = _ACGT[(np.searchsorted(_ACGT, t[sub]) + rng.integers(1, 4, size=int(sub.sum()))) % 4]
        r = np.repeat(t, counts)
        if ins.any():
            # second copy of every inserted base becomes a random base
            idx = np.cumsum(counts)[ins] - 1
            r[idx] = _ACGT[rng.integers(0, 4, size=idx.size)]
        pieces.append(r)
        offsets.append(offsets[-1] + r.size)
    return np.concatenate(pieces) if pieces else np.empty(0, np.uint8), np.array(offsets, dtype=np.uint64)


def write_fastq_fixed(path, bases, read_len, chunk=1 << 20):
    """Vectorised FASTQ writer for fixed-length reads (header '@r%09d'); fast enough for 10M-read files."""
    n = bases.size // read_len
    with open(path, "wb") as fh:
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            rec = np.empty((m, 12 + read_len + 1 + 2 + read_len + 1), dtype=np.uint8)
            rec[:, 0:2] = np.frombuffer(b"@r", np.uint8)
            idx = np.arange(lo, lo + m)
            for d in range(9):
                rec[:, 10 - d] = 48 + (idx // 10 ** d) % 10
            rec[:, 11] = 10
            rec[:, 12:12 + read_len] = bases[lo * read_len:(lo + m) * read_len].reshape(m, read_len)
            rec[:, 12 + read_len] = 10
            rec[:, 13 + read_len] = ord("+")
            rec[:, 14 + read_len] = 10
            rec[:, 15 + read_len:15 + 2 * read_len] = ord("I")
            rec[:, 15 + 2 * read_len] = 10
            fh.write(rec.tobytes())


def write_fastq(path, bases, offsets, gz=False):
    import gzip
    op = gzip.open if gz else open
    with op(path, "wb") as fh:
        for i in range(len(offsets) - 1):
            s = bases[int(offsets[i]):int(offsets[i + 1])].tobytes()
            fh.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")


# ---- an mtb-like panel built from a real index directory (genes.fa + panel.bcf) -------------------------------
def _fill_segment(rng, seg, fill_every, margin=12, max_alts=4, nested_frac=0.1):
    """random SNP bubbles about every `fill_every` bases inside a stretch of backbone that holds no panel site (SURVEY.md
    section 8d: "else seeded random SNP bubbles at 1 per 60 bp, <= 4 alts, 10 % nested depth-2"); the first-allele path stays
    the backbone.  Returns a segment list [str, Site, str, ...]."""
    out, cur = [], 0
    pos = margin + int(rng.integers(0, fill_every))
    while pos + 1 + margin <= len(seg):
        ref = seg[pos]
        others = "ACGT".replace(ref, "")
        n_alts = int(rng.integers(1, min(max_alts, 3) + 1))
        alleles = [[ref]] + [[others[i]] for i in rng.permutation(3)[:n_alts]]
        if rng.random() < nested_frac:
            inner_ref = random_seq(rng, 1)
            inner_alt = "ACGT".replace(inner_ref, "")[int(rng.integers(0, 3))]
            alleles.append([random_seq(rng, int(rng.integers(0, 6))), Site([[inner_ref], [inner_alt]]), random_seq(rng, int(rng.integers(2, 9)))])
        out.append(seg[cur:pos])
        out.append(Site(alleles))
        cur = pos + 1
        pos = cur + int(rng.integers(max(8, fill_every // 2), fill_every * 3 // 2 + 1))
    out.append(seg[cur:])
    return out


def panel_from_index_dir(index_dir, max_alts=4, min_gap=2, lead=30, fill_every=0, seed=20230308):
    """PRGs whose first-allele path is the genes.fa sequence and whose sites are the (non-overlapping) records of
    panel.bcf.  fill_every > 0 adds seeded random SNP bubbles at that spacing in the stretches the
    panel leaves empty (the 8d "mtb-like" index: fill_every=60).  Returns (Panel, sites) where
    sites[gene] = [(record id, pos, ref, alts)] lists the panel sites in order."""
    import os
    from .bcf_lite import read_bcf
    rng = np.random.default_rng(seed)
    genes = []
    name, seq = None, []
    for line in open(os.path.join(index_dir, "genes.fa")):
        if line.startswith(">"):
            if name:
                genes.append((name, "".join(seq).upper()))
            name, seq = line[1:].split()[0], []
        else:
            seq.append(line.strip())
    if name:
        genes.append((name, "".join(seq).upper()))
    by_gene = {}
    for r in read_bcf(os.path.join(index_dir, "panel.bcf")):
        by_gene.setdefault(r["chrom"], []).append(r)
    names, trees, sites = [], [], {}
    for gname, gseq in genes:
        recs = sorted(by_gene.get(gname, []), key=lambda r: (r["pos"], len(r["ref"])))
        segs, cur, chosen = [], 0, []
        for r in recs:
            if r["pos"] < max(cur + min_gap, lead) or r["pos"] + len(r["ref"]) > len(gseq) - lead:
                continue
            if gseq[r["pos"]:r["pos"] + len(r["ref"])] != r["ref"] or len(r["ref"]) > 12:
                continue
            alts = [a for a in dict.fromkeys(r["alts"]) if a != r["ref"] and set(a) <= set("ACGT")][:max_alts]
            if not alts:
                continue
            segs.append(gseq[cur:r["pos"]])
            segs.append(Site([[r["ref"]]] + [[a] for a in alts]))
            cur = r["pos"] + len(r["ref"])
            chosen.append((r["id"], r["pos"], r["ref"], alts))
        segs.append(gseq[cur:])
        if fill_every:
            filled = []
            for i, seg in enumerate(segs):
                if isinstance(seg, str) and len(seg) >= fill_every:
                    # (keep `lead` bases clear at both ends of the gene, as for the panel sites)
                    head = lead if i == 0 else 0
                    tail = lead if i == len(segs) - 1 else 0
                    body = seg[head:len(seg) - tail] if tail else seg[head:]
                    parts = _fill_segment(rng, body, fill_every)
                    parts[0] = seg[:head] + parts[0]
                    parts[-1] = parts[-1] + (seg[len(seg) - tail:] if tail else "")
                    filled.extend(parts)
                else:
                    filled.append(seg)
            segs = filled
        names.append(gname)
        trees.append(segs)
        sites[gname] = chosen
    return Panel(names, trees), sites


def mtb_8d_panel(index_dir=None):
    """The "mtb-like" index of SURVEY.md section 8d: backbone = the reference's test genes.fa (18 genes, 30,355 bp with
    padding 100), sites = the panel.bcf records that fit without overlapping + seeded random SNP bubbles at 1 per 60 bp
    elsewhere, <= 4 alts, 10 % nested; k = 15, w = 11, seed 20230308.  The index files are the committed copies under
    drprg_amd/data/mtb_8d/ (genes.fa and panel.bcf: data files of /root/reference/tests/cases/predict/, the same bytes as under
    tests/golden/downstream/ -- the package carries its own copy so that bench.py and smoke() do not depend on tests/)."""
    if index_dir is None:
        index_dir = MTB_8D_DIR
    return panel_from_index_dir(index_dir, fill_every=60)[0]


def mtb_scaled_panel(scale, index_dir=None, seed=20230308, site_every=24):
    """The 8d index grown `scale`-fold (round 3: how does the hot path behave between the 15 k k-mer nodes of the 8d index and
    the 620 k of the 500-locus one?): the 18 genes of mtb_8d_panel() plus (scale - 1) x 18 further loci `<gene>_x<j>` of the same
    padded lengths on seeded random backbones (65.6 % GC) with a site about every 24 bases (<= 4 alts, 10 % nested, 5 % indels),
    which gives them the k-mer node density of the 8d genes (~0.49 nodes per base: ~14.9 k nodes per 18 loci).  Reads sampled from
    HaplotypeGenomes of this panel hit it `scale` times as often as the 8d one (the loci cover scale x 0.7 % of the genome)."""
    base = mtb_8d_panel(index_dir)
    rng = np.random.default_rng(seed + 1000 * int(scale))
    names, trees = list(base.names), list(base.trees)
    for j in range(1, int(scale)):
        for n, L in MTB_LOCI:
            names.append(f"{n}_x{j}")
            trees.append(make_locus(rng, L, site_every=site_every, indel_frac=0.05))
    return Panel(names, trees)


def haplotype_with(segs, choose):
    """sequence of a locus where site number i (in order) takes allele choose(i) (0 = reference allele)"""
    s, i = "", 0
    for seg in segs:
        if isinstance(seg, str):
            s += seg
        else:
            s += "".join(x for x in seg.alleles[choose(i)] if isinstance(x, str))
            i += 1
    return s
