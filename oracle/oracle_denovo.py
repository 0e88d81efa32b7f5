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


# ---- noisy long reads (no -I): the column vote -------------------------------------------------------------------------------------------
# A second statement of the product's rules for reads that carry their own errors (drprg_amd/csrc/denovo.cpp column_consensus +
# assemble_candidate_regions with accurate_reads = false; DESIGN.md section 4 "Discover"), again in plain Python over whole reads:
# str.find for the anchors, a list-of-lists edit-distance table, dictionaries for the votes.  Like pile_up above it checks the
# IMPLEMENTATION of this build's rules; pandora's local assembly is not restated anywhere.
# Rules: up to three consecutive `anchor`-mers of consensus on either side of the padded region are searched exactly (either
# orientation); a read votes once per (region, orientation) with the innermost pair it holds (smallest jl + jr; among equals the first
# found scanning the read's anchor hits in position order, left anchor outermost loop) whose spelled length is within max_len_change of
# the expected one; a string with an N does not vote.  Every voting string is aligned globally at unit costs to the consensus slice
# between ITS anchors; the traceback prefers the diagonal, then a deletion, then an insertion (gaps end up as far left as the costs
# allow).  A column of the region collects votes A / C / G / T / deleted, the gap in front of a column the inserted strings.  With
# need = max(min_support, ceil(min_fraction * spanning)): an insertion is made where >= need reads insert something (its length the most
# frequent one -- the shortest among equally frequent --, its bases the per-place majority -- A < C < G < T on ties -- of the strings
# of that length); a column takes its most voted state (the consensus base wins ties, then the order A C G T deleted) if that state is the
# consensus base or has >= need votes.  The result is aligned to again, up to three rounds, until it stops changing.  Reported support =
# reads whose string is closer (edit distance) to the new allele than to the consensus, both extended by the skipped anchors.
def _align_votes(ref, s, shift, R, weight, col, ins):
    E, S = len(ref), len(s)
    D = [[0] * (S + 1) for _ in range(E + 1)]
    for i in range(E + 1):
        D[i][0] = i
    for j in range(S + 1):
        D[0][j] = j
    for i in range(1, E + 1):
        for j in range(1, S + 1):
            D[i][j] = min(D[i - 1][j - 1] + (ref[i - 1] != s[j - 1]), D[i - 1][j] + 1, D[i][j - 1] + 1)
    i, j, pending = E, S, []

    def flush(before):
        if pending:
            if shift <= before <= shift + R:
                key = "".join(reversed(pending))
                ins[before - shift][key] = ins[before - shift].get(key, 0) + weight
            del pending[:]

    while i > 0 or j > 0:
        if i > 0 and j > 0 and D[i][j] == D[i - 1][j - 1] + (ref[i - 1] != s[j - 1]):
            flush(i)
            if s[j - 1] in "ACGT" and shift <= i - 1 < shift + R:
                col[i - 1 - shift]["ACGT".index(s[j - 1])] += weight
            i, j = i - 1, j - 1
        elif i > 0 and D[i][j] == D[i - 1][j] + 1:
            flush(i)
            if shift <= i - 1 < shift + R:
                col[i - 1 - shift][4] += weight
            i -= 1
        else:
            pending.append(s[j - 1])
            j -= 1
    flush(0)
    return D[E][S]


def _edit(a, b):
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j - 1] + (a[i - 1] != b[j - 1]), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[len(b)]


def column_vote(consensus, regions, reads, anchor=15, min_support=4, min_fraction=0.6, max_len_change=30, max_anchors=3):
    """Same arguments and result layout as pile_up: [(locus, pos0, ref, alt, support, spanning)]."""
    import math
    out = []
    for locus, start, end in regions:
        cons = consensus[locus]
        if start < anchor or end + anchor > len(cons):
            continue
        nl, nr = min(max_anchors, start // anchor), min(max_anchors, (len(cons) - end) // anchor)
        left = [cons[start - anchor * (j + 1):start - anchor * j] for j in range(nl)]    # left[0] is next to the region
        right = [cons[end + anchor * j:end + anchor * (j + 1)] for j in range(nr)]
        core0 = cons[start:end]
        votes = {}
        for read in reads:
            read = read.upper()
            if len(read) < 2 * anchor:
                continue
            for reverse in (False, True):
                hits = []  # (position, side, j): side 0 = a left anchor of the consensus, 1 = a right one
                for side, lst in ((0, left), (1, right)):
                    for j, a in enumerate(lst):
                        pat = revcomp(a) if reverse else a
                        hits += [(p, side, j) for p in _occurrences(read, pat)]
                hits.sort()
                best = None
                for x in hits:
                    for y in hits:
                        first_ok = (x[1] == 1 and y[1] == 0) if reverse else (x[1] == 0 and y[1] == 1)
                        if not first_ok or y[0] < x[0] + anchor:
                            continue
                        jl, jr = (y[2], x[2]) if reverse else (x[2], y[2])
                        got, want = y[0] - (x[0] + anchor), (end - start) + anchor * (jl + jr)
                        if got > want + max_len_change or got + max_len_change < want:
                            continue
                        if best is None or jl + jr < best[2] + best[3]:
                            best = (x[0] + anchor, got, jl, jr)
                if best is None:
                    continue
                s = read[best[0]:best[0] + best[1]]
                if reverse:
                    s = revcomp(s)
                if any(c not in "ACGT" for c in s):
                    continue
                votes[(best[2], best[3], s)] = votes.get((best[2], best[3], s), 0) + 1
        spanning = sum(votes.values())
        if spanning < min_support:
            continue
        need = max(min_support, int(math.ceil(min_fraction * spanning)))
        ext = lambda core, jl, jr: cons[start - anchor * jl:start] + core + cons[end:end + anchor * jr]
        allele = core0
        for _ in range(3):
            R = len(allele)
            col = [[0, 0, 0, 0, 0] for _ in range(R)]
            ins = [dict() for _ in range(R + 1)]
            for (jl, jr, s), n in votes.items():
                _align_votes(ext(allele, jl, jr), s, anchor * jl, R, n, col, ins)
            nxt = []
            for i in range(R + 1):
                if sum(ins[i].values()) >= need:
                    by_len = {}
                    for t, n in ins[i].items():
                        by_len[len(t)] = by_len.get(len(t), 0) + n
                    top = max(by_len.values())
                    ln = min(l for l, n in by_len.items() if n == top)
                    for p in range(ln):
                        cnt = [0, 0, 0, 0]
                        for t, n in ins[i].items():
                            if len(t) == ln and t[p] in "ACGT":
                                cnt["ACGT".index(t[p])] += n
                        nxt.append("ACGT"[cnt.index(max(cnt))])
                if i == R:
                    break
                rc = "ACGT".index(allele[i])
                arg = rc
                for c in range(5):
                    if col[i][c] > col[i][arg]:
                        arg = c
                if arg != rc and col[i][arg] < need:
                    arg = rc
                if arg < 4:
                    nxt.append("ACGT"[arg])
            nxt = "".join(nxt)
            if nxt == allele:
                break
            allele = nxt
        if allele == core0:
            continue
        support = sum(n for (jl, jr, s), n in votes.items() if _edit(s, ext(allele, jl, jr)) < _edit(s, ext(core0, jl, jr)))
        pre = 0
        while pre < len(core0) and pre < len(allele) and core0[pre] == allele[pre]:
            pre += 1
        suf = 0
        while suf < len(core0) - pre and suf < len(allele) - pre and core0[len(core0) - 1 - suf] == allele[len(allele) - 1 - suf]:
            suf += 1
        out.append((locus, start + pre, core0[pre:len(core0) - suf], allele[pre:len(allele) - suf], support, spanning))
    return sorted(out, key=lambda v: (v[0], v[1]))


# ---- local assembly (round 4; accurate reads) ---------------------------------------------------------------------------------------------
# A second statement of the product's de Bruijn assembly of a candidate region (drprg_amd/csrc/denovo.cpp assemble_region +
# assemble_candidate_regions with DRPRG_HIP_DENOVO=dbg), in plain Python: str.find-free set / dict code over whole reads, recursion instead
# of the product's explicit stack.  It follows what pandora discover does as far as remembered [UPSTREAM-MEMORY: de Bruijn graph over the
# reads of a region, k = 15, nodes seen at least twice, depth-first paths from a start k-mer to an end k-mer taken from the region's
# flanks, outermost pair first] -- pandora's source is not in the reference tree, so like everything else in this file it checks the
# IMPLEMENTATION of this build's rules.
# Rules: slice = the `anchor` consensus bases before the padded region + the region + the `anchor` bases behind it.  A read (>= anchor
# bases) joins the region's pile once per orientation in which it shares an anchor-mer with the slice (the reverse complement of the read
# joins when it shares one with the slice's reverse complement, i.e. reads are piled in the consensus' orientation).  Nodes: the
# anchor-mers of the piled reads (no N) seen >= min_dp times.  Start k-mers: slice offsets 0 .. flank_l - k, end k-mers: offsets
# len - k - (0 .. flank_r - k), flank = anchor + padding on that side (region start -> first low-coverage base).  For the start k-mers in
# that order and, per start, the end k-mers in that order (a k-mer must be a node, the end must lie behind the start): all paths from
# start to end of total length within max_len_change of the consensus stretch between them; more than max_paths paths, or more than
# 200000 extension attempts: this pair is skipped; the first pair with at least one path decides.  A path that spells the consensus is
# dropped, as is one whose k-mers are all consensus k-mers; support = the smallest count among its non-consensus k-mers.  Alleles with
# support >= max(min_support, ceil(best / 10)) are reported, trimmed against the slice, best supported first (ties: by sequence);
# spanning = the largest count among the start / end k-mers.
def local_assembly(consensus, regions, reads, anchor=15, min_dp=2, min_support=3, max_len_change=30, max_paths=25, max_steps=200000):
    """regions: [(locus, start, end, low_start, low_end)].  Returns [(locus, pos0, ref, alt, support, spanning)] in region order, the
    alleles of a region best supported first."""
    K = anchor
    out = []
    for locus, start, end, low_start, low_end in regions:
        cons = consensus[locus]
        if start < K or end + K > len(cons):
            continue
        sl = cons[start - K:end + K]
        rc_sl = revcomp(sl)
        fwd_k = {sl[i:i + K] for i in range(len(sl) - K + 1)}
        rev_k = {rc_sl[i:i + K] for i in range(len(rc_sl) - K + 1)}
        pile = []
        for read in reads:
            read = read.upper()
            if len(read) < K:
                continue
            ks = {read[i:i + K] for i in range(len(read) - K + 1)}
            ks = {k for k in ks if all(c in "ACGT" for c in k)}
            if ks & fwd_k:
                pile.append(read)
            if ks & rev_k:
                pile.append(revcomp(read))
        count = {}
        for r in pile:
            for i in range(len(r) - K + 1):
                k = r[i:i + K]
                if all(c in "ACGT" for c in k):
                    count[k] = count.get(k, 0) + 1
        node = lambda k: count.get(k, 0) >= min_dp
        flank_l, flank_r = K + (low_start - start), K + (end - low_end)
        starts = list(range(0, max(0, flank_l - K + 1)))
        ends = [len(sl) - K - i for i in range(0, max(0, flank_r - K + 1))]
        depth = max([count.get(sl[i:i + K], 0) for i in starts + ends] or [0])
        found = None
        for si in starts:
            if found is not None:
                break
            if not node(sl[si:si + K]):
                continue
            for eo in ends:
                if eo <= si or not node(sl[eo:eo + K]):
                    continue
                want = eo + K - si
                target, paths, steps = sl[eo:eo + K], [], [0]

                class TooMany(Exception):
                    pass

                def walk(kmer, tail):
                    if kmer == target and K + len(tail) >= max(K, want - max_len_change):
                        paths.append(tail)
                        if len(paths) > max_paths:
                            raise TooMany()
                    if K + len(tail) >= want + max_len_change:
                        return
                    for b in "ACGT":
                        steps[0] += 1
                        if steps[0] > max_steps:
                            raise TooMany()
                        nxt = kmer[1:] + b
                        if node(nxt):
                            walk(nxt, tail + b)

                try:
                    import sys
                    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4 * (want + max_len_change) + 200))
                    walk(sl[si:si + K], "")
                except TooMany:
                    continue
                if not paths:
                    continue
                found = {}
                for tail in paths:
                    spelled = sl[si:si + K] + tail
                    whole = sl[:si] + spelled + sl[eo + K:]
                    if whole == sl:
                        continue
                    novel = [count.get(spelled[i:i + K], 0) for i in range(len(spelled) - K + 1) if spelled[i:i + K] not in fwd_k]
                    if not novel:
                        continue
                    found[whole] = max(found.get(whole, 0), min(novel))
                break
        if not found:
            continue
        ranked = sorted(found.items(), key=lambda kv: (-kv[1], kv[0]))
        need = max(min_support, (ranked[0][1] + 9) // 10)
        for whole, support in ranked:
            if support < need:
                break
            pre = 0
            while pre < len(sl) and pre < len(whole) and sl[pre] == whole[pre]:
                pre += 1
            suf = 0
            while suf < len(sl) - pre and suf < len(whole) - pre and sl[len(sl) - 1 - suf] == whole[len(whole) - 1 - suf]:
                suf += 1
            out.append((locus, start - K + pre, sl[pre:len(sl) - suf], whole[pre:len(whole) - suf], support, depth))
    return out
