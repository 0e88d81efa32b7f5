#!/usr/bin/env python3
"""bench.py -- reads/s of the drprg predict hot path on MI355X (metric of BASELINE.json).

A "step" is one pass of the hot path (sketch + probe + cluster + coverage accumulation, then the
sum-reduce of the coverage vector across ranks) over one batch of synthetic reads that is already
resident in HBM.  At N=1 the workload is BASELINE.json configs[1]: 10M synthetic 150 bp Illumina reads
against the mtb-like PRG index of SURVEY.md section 8d (backbone = the reference's test genes.fa, sites =
its panel.bcf records + seeded random bubbles; 18 loci, k=15, w=11).  With N>1 every rank maps its own
shard of BASELINE.json configs[3] -- 200M reads over 8 GPUs = 25M reads per GPU; N = 2 and 4 map the same 25M-read shard
per GPU (weak-scaling points of configs[3]) -- and the only data-path collective is ONE RCCL all-reduce of the u32 vector
[coverage | reads per PRG] per step: `--comm native` (the default with RCCL) issues it through the C ABI
(drprg_hip_comm_unique_id / comm_init_rank / drprg_hip_allreduce: what a Rust host would call; rank 0's id travels over the
torch.distributed group the launcher set up), `--comm torch` through torch.distributed's own RCCL binding.

`--gpus N` with N > 1 and no launcher in the environment starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, spawned before this process touches
the GPU) and relays rank 0's line and the exit code; it refuses to run with fewer than N devices.

Steps are queued back to back (drprg_hip_map_device_async: the read-back a batch ends with is looked at
while the next batch runs); DRPRG_BENCH_SYNC=1 makes the host wait for every batch instead.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (sketch_filter_kernel for the
mtb-sized indexes, sketch_wave_kernel for the 500-locus one) against HBM bandwidth using the
algorithmic bytes of SURVEY.md section 8d, with its duration measured live with HIP events on the
launch stream; `cpu_baseline` times the CPU oracle (oracle/oracle.c + oracle_index.c, a scalar port) on a
bounded sample of the same workload, on one thread and on the host's cores (up to 64 threads).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBS = 6290.0  # same guide, line 36: 6.29 TB/s measured (float4 copy)


def gpu_sample_reads(torch, hap_pad, hap_lens, n_reads, read_len, seed, device, sub_rate=0.001, chunk=1 << 20):
    """Same distribution as drprg_amd.synth.sample_short_reads, generated on the device (plumbing only)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_hap, max_len = hap_pad.shape
    flat = hap_pad.reshape(-1)
    out = torch.empty(n_reads * read_len, dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device, dtype=torch.int64)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    for lo in range(0, n_reads, chunk):
        m = min(chunk, n_reads - lo)
        hap = torch.randint(0, n_hap, (m,), generator=g, device=device)
        span = (hap_lens[hap] - read_len).to(torch.float64)
        start = (torch.rand(m, generator=g, device=device, dtype=torch.float64) * span).to(torch.int64)
        idx = (hap * max_len + start)[:, None] + ar
        block = flat[idx]
        rev = torch.rand(m, generator=g, device=device) < 0.5
        rc = comp[block.flip(1).long()]
        block = torch.where(rev[:, None], rc, block)
        err = torch.rand(block.shape, generator=g, device=device) < sub_rate
        rnd = acgt[torch.randint(0, 4, block.shape, generator=g, device=device)]
        block = torch.where(err, rnd, block)
        out[lo * read_len:(lo + m) * read_len] = block.reshape(-1)
    offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=device) * read_len
    return out, offsets


def gpu_sample_long_reads(torch, hap_pad, hap_lens, n_reads, seed, device, mean_len=4000, sigma=0.5, min_len=500, max_len=50000,
                          err=0.05, chunk_bases=1 << 25):
    """Same distribution as drprg_amd.synth.sample_long_reads (lognormal lengths, 5 % errors split 40/30/30
    substitution/insertion/deletion), generated on the device chunk by chunk (plumbing only)."""
    import math
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_hap, max_hap = hap_pad.shape
    flat = hap_pad.reshape(-1)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    mu = math.log(mean_len) - sigma * sigma / 2
    z = torch.randn(n_reads, generator=g, device=device, dtype=torch.float64)
    lens = torch.exp(mu + sigma * z).clamp(min_len, max_len).to(torch.int64)
    hap = torch.randint(0, n_hap, (n_reads,), generator=g, device=device)
    lens = torch.minimum(lens, hap_lens[hap] - 1)
    start = (torch.rand(n_reads, generator=g, device=device, dtype=torch.float64) * (hap_lens[hap] - lens).to(torch.float64)).to(torch.int64)
    rev = torch.rand(n_reads, generator=g, device=device) < 0.5
    src_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    src_off[1:] = torch.cumsum(lens, 0)
    bounds = [0]  # chunks of whole reads holding ~chunk_bases source bases
    so = src_off.cpu().numpy()
    while bounds[-1] < n_reads:
        nxt = int(np.searchsorted(so, so[bounds[-1]] + chunk_bases, side="right")) - 1
        bounds.append(min(n_reads, max(nxt, bounds[-1] + 1)))
    pieces, out_lens = [], torch.empty(n_reads, dtype=torch.int64, device=device)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ln = lens[lo:hi]
        m = int(ln.sum().item())
        rid = torch.repeat_interleave(torch.arange(hi - lo, device=device), ln)
        i_in = torch.arange(m, device=device) - (src_off[lo:hi] - src_off[lo])[rid]
        r = rev[lo:hi][rid]
        src = torch.where(r, start[lo:hi][rid] + ln[rid] - 1 - i_in, start[lo:hi][rid] + i_in)
        t = flat[hap[lo:hi][rid] * max_hap + src]
        t = torch.where(r, comp[t.long()], t)
        u = torch.rand(m, generator=g, device=device)
        counts = torch.ones(m, dtype=torch.int64, device=device)
        counts[u < err * 0.3] = 0
        ins = (u >= err * 0.3) & (u < err * 0.6)
        counts[ins] = 2
        sub = (u >= err * 0.6) & (u < err)
        rnd = acgt[torch.randint(0, 4, (m,), generator=g, device=device)]
        t = torch.where(sub, rnd, t)
        out = torch.repeat_interleave(t, counts)
        idx = torch.cumsum(counts, 0)[ins] - 1
        out[idx] = acgt[torch.randint(0, 4, (int(idx.numel()),), generator=g, device=device)]
        out_lens[lo:hi] = torch.zeros(hi - lo, dtype=torch.int64, device=device).index_add_(0, rid, counts)
        pieces.append(out)
        del rid, i_in, r, src, t, u, counts, ins, sub, rnd, idx
    offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(out_lens, 0)
    return torch.cat(pieces), offsets


CONFIGS3_TOTAL_READS = 200_000_000  # BASELINE.json configs[3]: 8 x MI355X, 200M synthetic 150 bp reads sharded across the GPUs
CONFIGS3_GPUS = 8

WORKLOADS = {
    # name: (BASELINE.json config, description, default reads per GPU, illumina, panel)
    "mtb": ("configs[1]", "10M synthetic 150 bp Illumina reads vs mtb-like PRG index (reference genes.fa backbone + panel.bcf sites, "
            "SURVEY 8d)", 10_000_000, True, "mtb_8d"),
    "mtb-random": ("configs[1]", "10M synthetic 150 bp Illumina reads vs mtb-like PRG index (random backbone with the 18 loci's lengths, "
                   "1 site / 60 bp)", 10_000_000, True, "mtb_like"),
    "nanopore": ("configs[2]", "2M synthetic Nanopore reads (mean 4 kb, 5% error) vs mtb-like PRG index (SURVEY 8d)", 2_000_000, False,
                 "mtb_8d"),
    "big": ("configs[4]", "10M synthetic 150 bp reads vs 500-locus / 50k-variant synthetic PRG index", 10_000_000, True, "big"),
}
# the 8d index grown 2-32 fold (drprg_amd.synth.mtb_scaled_panel: 30 k ... 489 k k-mer nodes): how the hot path degrades between the
# 15 k nodes of the 8d index and the 620 k of the 500-locus one (not BASELINE configurations; profiles/r03 holds a line per size)
WORKLOADS["mtb-dense"] = ("configs[1] reads, denser index", "10M synthetic 150 bp Illumina reads vs the 8d genes with a seeded SNP bubble about every 8 "
                          "bases between the panel sites (36 k k-mer nodes on the same 30 kb: the index grows, the hits do not)", 10_000_000, True,
                          "mtb_dense")
for _s in (2, 4, 8, 16, 32):
    WORKLOADS[f"mtb-x{_s}"] = ("configs[1] reads, larger index", f"10M synthetic 150 bp Illumina reads vs the 8d mtb-like index grown {_s}-fold "
                               f"({18 * _s} loci)", 10_000_000, True, f"mtb_x{_s}")


# the source files of the dominant kernels: profiles/traffic.json and profiles/valu.json say which state of them their numbers were
# measured on (tools/make_profile_json.py holds the same table)
KERNEL_SOURCES = {
    "sketch_filter_kernel": ["sketch_filter.hip", "filter_common.h", "device_common.h", "kernels.h", "common.h"],
    "sketch_wave_kernel": ["sketch_wave.hip", "sketch_block.h", "device_common.h", "kernels.h", "common.h"],
    "sketch_probe_kernel": ["sketch_probe.hip", "device_common.h", "kernels.h", "common.h"],
}


def kernel_source_sha16(kernel):
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(kernel, []):
        h.update(open(os.path.join(ROOT, "drprg_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def profile_entry_stale(entry, kernel):
    """None if the replayed profile entry was measured on the kernel sources of this tree, else the reason it is dropped"""
    if not entry:
        return None
    if kernel not in KERNEL_SOURCES:
        return f"no source table for {kernel}"
    # (the profiles are of the default form of the kernel: a run that asks for another one -- A/B lines -- replays nothing)
    forced = [f"{k}={os.environ[k]}" for k in ("DRPRG_FILTER_STAGE2", "DRPRG_FILTER_FORM", "DRPRG_DIRECT_FORM", "DRPRG_FT_DEBUG") if os.environ.get(k)]
    if forced:
        return "this run asks for a form of the kernel the profiles were not taken on (" + ", ".join(forced) + ")"
    was = entry.get("source_sha16")
    now = kernel_source_sha16(kernel)
    if was is None:
        return f"the profile entry does not name the kernel sources it was measured on ({', '.join(KERNEL_SOURCES[kernel])} are now {now})"
    if was != now:
        return f"{', '.join(KERNEL_SOURCES[kernel])} changed since the profile was taken (sha256 {was} then, {now} now): run tools/run_profiles_r06.sh + tools/make_profile_json.py"
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def make_panel(synth, which):
    if which.startswith("mtb_x"):
        return synth.mtb_scaled_panel(int(which[5:]))
    if which == "mtb_dense":
        return synth.panel_from_index_dir(synth.MTB_8D_DIR, fill_every=8)[0]
    return {"mtb_8d": synth.mtb_8d_panel, "mtb_like": synth.mtb_like_panel, "big": synth.big_panel}[which]()


def spawn_ranks(n, argv):
    """`--gpus N` without a launcher: start N ranks as a child torch.distributed.run (one process per GPU, RCCL) and return its
    exit code.  Nothing in this process has initialised the GPU (torch.cuda.device_count() does not on this image)."""
    import socket
    import subprocess
    if os.environ.get("DRPRG_BENCH_BACKEND", "nccl") == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} device(s) are visible; refusing to run fewer ranks than asked for",
                  file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def map_range(ctx, bases, offsets, lo, hi, covg, prg_reads, stream, torch):
    """reads [lo, hi) of the batch through the hot path (a 16-byte aligned copy of that range when lo > 0)"""
    b0, b1 = int(offsets[lo].item()), int(offsets[hi].item())
    if lo == 0:
        sub_b, sub_o = bases, offsets
    else:
        sub_b = bases[b0:b1].clone()
        sub_o = (offsets[lo:hi + 1] - b0).contiguous()
    torch.cuda.synchronize()  # the copies above ran on torch's current stream, the hot path runs on `stream`
    ctx.map_device(sub_b.data_ptr(), sub_o.data_ptr(), hi - lo, b1 - b0, covg.data_ptr(), prg_reads.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()


def full_size_checks(torch, ctx, opts, bases, offsets, n_reads, covg, prg_reads, stream):
    """Size-independent parity properties at full size (the oracle cannot map the whole batch in seconds):
    (1) sharding invariance: coverage(whole shard) == coverage(first half) + coverage(second half), bit for bit;
    (2) the sequence in use (Bloom-prefiltered, or direct in its candidate form) and the direct kernel with the generic
    cluster pipeline (radix sort + cluster kernels) give the identical vector."""
    full = covg.clone()
    split = torch.zeros_like(covg)
    sp = torch.zeros_like(prg_reads)
    torch.cuda.synchronize()
    half = n_reads // 2
    map_range(ctx, bases, offsets, 0, half, split, sp, stream, torch)
    map_range(ctx, bases, offsets, half, n_reads, split, sp, stream, torch)
    shard_invariant = bool(torch.equal(full, split))
    kernels_agree = None
    if ctx.counters().get("kernel") in (2, 3):  # against the direct kernel + generic cluster pipeline (sort, cluster kernels)
        ctx.set_opts(kernel=1, **opts)
        direct = torch.zeros_like(covg)
        map_range(ctx, bases, offsets, 0, n_reads, direct, sp, stream, torch)
        kernels_agree = bool(torch.equal(full, direct))
        ctx.set_opts(kernel=int(os.environ.get("DRPRG_BENCH_KERNEL", "0")), **opts)
    return shard_invariant, kernels_agree


def write_bgzf(path, data, block=65280, level=1):
    """bgzip's container (gzip members of <= 64 KB with a 'BC' extra field), written with zlib only"""
    import struct
    import zlib
    with open(path, "wb") as fh:
        for lo in list(range(0, len(data), block)) + [None]:
            chunk = data[lo:lo + block] if lo is not None else b""
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            fh.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1))
            fh.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def e2e_leg(torch, ctx, synth, bases, n_reads, read_len, covg):
    """End to end (SURVEY 8d "end-to-end and kernel-only"): the same reads as FASTQ text in the page cache (/dev/shm) -> parse
    on host threads -> pinned blocks -> PCIe -> kernels -> coverage, through drprg_hip_map_fastx; outside the timed region.
    Plain text: the whole batch, and its coverage must equal the HBM-resident result.  gzip (BGZF as bgzip writes it, and one
    plain member as `gzip` writes it): the first 2 M reads."""
    import gzip
    import shutil
    # parser / inflate threads: as many as the process may actually run at once -- the CPU quota of its cgroup when there is one (the
    # pool's boxes show 256 hardware threads under a quota of 16 CPUs: 32 parser threads then take twice as long as 16, measured
    # 92-110 against 43-58 ms per 10 M reads, profiles/r04/e2e_threads.txt) -- at most 32
    threads = max(1, min(os.cpu_count() or 1, 32))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    d = tempfile.mkdtemp(prefix="drprg_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    res = {"threads": threads, "input": "FASTQ text in the page cache; wall time of drprg_hip_map_fastx (parse + pin + PCIe + kernels)"}
    try:
        host = bases.cpu().numpy()
        fq = os.path.join(d, "reads.fq")
        synth.write_fastq_fixed(fq, host, read_len)
        ctx.set_threads(threads)
        want = covg.cpu().numpy().view(np.uint32)

        def run(path, n, reps=3):
            """median of `reps` runs (the host of the GPU box is shared: single runs of the same file spread by a factor of two)"""
            times = []
            for _ in range(reps):
                ctx.reset()
                t0 = time.perf_counter()
                ctx.map_fastx(path)
                times.append(time.perf_counter() - t0)
            got = ctx.coverage()[0]
            dt = sorted(times)[len(times) // 2]
            return {"reads": n, "seconds": dt, "best_seconds": min(times), "runs": reps, "reads_per_s": n / dt,
                    "file_GB_per_s": os.path.getsize(path) / dt / 1e9}, got

        run(fq, n_reads, 1)  # (first pass: pinned blocks and workspaces are allocated)
        res["plain"], got = run(fq, n_reads)
        res["plain"]["coverage_equals_hbm_resident_run"] = bool(np.array_equal(got, want))
        # the same file with the parser threads packing the bases to 2 bits (drprg_hip_set_input_format): a quarter of the bytes to
        # page-lock and to move over PCIe
        ctx.set_input_format(True)
        run(fq, n_reads, 1)
        res["plain_packed"], got = run(fq, n_reads)
        res["plain_packed"]["coverage_equals_hbm_resident_run"] = bool(np.array_equal(got, want))
        res["plain_packed"]["pcie_floor_s"] = ((int(host.size) + 15) // 16 * 4 + 8 * (n_reads + 1)) / 63e9
        ctx.set_input_format(False)
        # what the link alone would take for the bases + offsets of this batch (PCIe Gen5 x16: 63 GB/s spec, MI355X_MICROARCH.md):
        # the gap to `seconds` is host work (file reads, parse, hand-over) that the copies do not hide
        res["plain"]["pcie_floor_s"] = (int(host.size) + 8 * (n_reads + 1)) / 63e9
        n_gz = min(n_reads, 2_000_000)
        rec = os.path.getsize(fq) // n_reads
        text = open(fq, "rb").read(rec * n_gz)
        ref = None
        def write_gzip(p):
            with gzip.open(p, "wb", compresslevel=1) as fh:
                fh.write(text)

        for name, writer in (("bgzf", lambda p: write_bgzf(p, text)), ("gzip", write_gzip)):
            p = os.path.join(d, f"reads.{name}.fq.gz")
            writer(p)
            res[name], got = run(p, n_gz)
            if ref is None:
                sub = os.path.join(d, "sub.fq")
                open(sub, "wb").write(text)
                ref = run(sub, n_gz, 1)[1]
            res[name]["coverage_equals_plain_text_run"] = bool(np.array_equal(got, ref))
            res[name]["how"] = ("BGZF members located from their headers, inflated in parallel by libdeflate" if name == "bgzf" else
                                "one plain gzip member inflated by all threads: chunks entered at block boundaries found in the compressed data "
                                "(csrc/pgunzip.cpp)")
    finally:
        shutil.rmtree(d, ignore_errors=True)
        ctx.reset()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="mtb", choices=sorted(WORKLOADS),
                    help="mtb = configs[1] (the bench line); mtb-random = the same reads against a random-backbone panel; "
                         "nanopore = configs[2]; big = configs[4]'s index; mtb-xN = the 8d index grown N-fold")
    ap.add_argument("--reads-per-gpu", type=int, default=0, help="0 = the workload's size (N > 1 with the mtb workload: 25M, the per-GPU "
                                                                 "shard of configs[3])")
    ap.add_argument("--input", default="ascii", choices=("ascii", "packed"),
                    help="format of the HBM-resident batch: ascii (one byte per base; the headline) or packed (2 bits per base, "
                         "include/drprg_hip.h 'packed reads': the batch is packed on the device before the timed region).  Either way "
                         "the roofline prices the SURVEY 8d bytes, L + 8 per read (packing is an optimisation, not a change of work)")
    ap.add_argument("--comm", default="auto", choices=("auto", "native", "torch"),
                    help="N > 1: who issues the all-reduce.  native = the C ABI (drprg_hip_comm_* / drprg_hip_allreduce, ONE "
                         "ncclAllReduce of [coverage | reads per PRG]); torch = torch.distributed.all_reduce on the same buffer; "
                         "auto = native with the nccl backend, torch otherwise")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads timed on the CPU oracle (0 = skip, -1 = ~10 s worth)")
    ap.add_argument("--e2e", type=int, default=1, help="1: add the end-to-end leg (FASTQ text in /dev/shm -> coverage through map_fastx, "
                                                       "plain and gzip) to the JSON line at N=1; 0: skip")
    ap.add_argument("--spinup-ms", type=float, default=float(os.environ.get("DRPRG_BENCH_SPINUP_MS", "600")),
                    help="untimed mapping of the same batch for this long BEFORE the warm-up steps, so that the timed region runs on a device at "
                         "its clocks (0: none; the line's cold_start leg is the same command on the device as the preparation leaves it)")
    ap.add_argument("--no-checks", action="store_true",
                    help="skip the full-size property checks after the timed region (profiling runs: every launch is then a "
                         "timed full-size one, so rocprofv3 per-kernel averages compare directly with avg_launch_ms)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # test hook for boxes with fewer GPUs than ranks: DRPRG_BENCH_BACKEND=gloo puts every rank on GPU 0 and reduces
    # through gloo, which exercises the N > 1 control flow of this file (not a benchmark configuration)
    backend = os.environ.get("DRPRG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from datetime import timedelta
        limit = timedelta(minutes=5)  # (a rank that never arrives fails the run in minutes, not in torch's default half hour)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=limit)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=limit)

    from drprg_amd import Context, synth

    W, K = 11, 15
    cfg_name, cfg_desc, default_reads, illumina, which_panel = WORKLOADS[args.workload]
    panel = make_panel(synth, which_panel)
    tmp = tempfile.mkdtemp(prefix=f"drprg_bench_r{rank}_")
    prg = os.path.join(tmp, "dr.prg")
    panel.write(prg, os.path.join(tmp, "genes.fa"))
    ctx = Context(prg, W, K, device=local_rank, from_files=False, threads=8)
    opts = dict(illumina=illumina, min_cluster_size=10, genome_size=synth.MTB_GENOME_SIZE)
    forced_kernel = int(os.environ.get("DRPRG_BENCH_KERNEL", "0"))  # experiments: 3 = the direct sequence where auto picks the filtered one
    ctx.set_opts(kernel=forced_kernel, **opts)

    # synthetic reads, generated on the device; every rank samples a different shard (seed + rank)
    genomes = synth.HaplotypeGenomes(panel, n_hap=8)
    hap_pad = torch.from_numpy(genomes.padded()).to(device)
    hap_lens = torch.from_numpy(genomes.lens).to(device)
    n_reads = args.reads_per_gpu or default_reads
    if world > 1 and args.workload == "mtb" and not args.reads_per_gpu:
        # BASELINE.json configs[3]: 200M reads sharded over 8 GPUs = 25M per GPU; N = 2 / 4 map the same per-GPU shard (weak scaling)
        n_reads = CONFIGS3_TOTAL_READS // CONFIGS3_GPUS
        cfg_name = "configs[3]" if world == CONFIGS3_GPUS else f"configs[3] weak-scaling point ({world} of {CONFIGS3_GPUS} GPUs, the same {n_reads // 1_000_000}M-read shard per GPU)"
        cfg_desc = (f"{world} x MI355X, {n_reads * world // 1_000_000}M synthetic 150 bp Illumina reads sharded across the GPUs ({n_reads // 1_000_000}M per GPU), "
                    "RCCL-reduced coverage vs mtb-like PRG index (SURVEY 8d)")
    if args.workload == "nanopore":
        bases, offsets = gpu_sample_long_reads(torch, hap_pad, hap_lens, n_reads, 3 + rank, device)
    else:
        bases, offsets = gpu_sample_reads(torch, hap_pad, hap_lens, n_reads, args.read_len, 2 + rank, device)
    n_bases = int(bases.numel())
    del hap_pad
    torch.cuda.synchronize()  # (torch made the batch on ITS stream; the context's calls below run on theirs)
    torch.cuda.empty_cache()
    packed = args.input == "packed"
    d_words = d_npos = None
    n_npos = 0
    if packed:  # the same batch, 2 bits per base (outside the timed region: what a packing ingest hands over)
        d_words = torch.zeros((n_bases + 15) // 16 + 4, dtype=torch.int32, device=device)
        d_npos = torch.zeros(1 << 16, dtype=torch.int64, device=device)
        torch.cuda.synchronize()  # (the two fills above are on torch's stream, the packing on the context's)
        n_npos = ctx.pack_device(bases.data_ptr(), n_bases, d_words.data_ptr(), d_npos.data_ptr(), d_npos.numel())
    # the reduced vector: per-node coverage and per-PRG cluster counts in one buffer (one memset, one all-reduce).  With
    # N > 1 two buffers alternate: the all-reduce of step i runs on RCCL's stream while step i+1 maps into the other
    # buffer; a buffer is only zeroed again once its reduce has finished (every step's collective completes inside the
    # timed region: drain() before the closing barrier).  DRPRG_BENCH_SYNC_REDUCE=1: one buffer, the stream waits for
    # every reduce before the next step starts.
    n_acc = 2 * ctx.n_knodes + ctx.n_prgs
    overlap = world > 1 and os.environ.get("DRPRG_BENCH_SYNC_REDUCE", "0") in ("", "0")
    # Steps are queued without the host waiting for them (drprg_hip_map_device_async): the read-back a batch ends with is looked
    # at while the next batch runs, so the device never idles between steps.  A batch's accumulator must stay untouched until
    # the call after the next one, hence buffers in rotation: 2 at N = 1; 3 at N > 1, where the all-reduce of batch i is issued
    # once batch i is known to be complete -- right after batch i+1 has been queued -- and runs on RCCL's stream while batch
    # i+1 / i+2 map into the other buffers (every step's collective completes inside the timed region: drain() before the
    # closing barrier).  DRPRG_BENCH_SYNC_REDUCE=1: one buffer, synchronous calls, the stream waits for every reduce.
    deferred = (world == 1 or overlap) and os.environ.get("DRPRG_BENCH_SYNC", "0") in ("", "0")  # DRPRG_BENCH_SYNC=1: the host waits for every batch
    accs = [torch.zeros(n_acc, dtype=torch.int32, device=device) for _ in range((3 if world > 1 else 2) if deferred else 1)]
    pending = [None] * len(accs)
    stream = torch.cuda.Stream(device)  # the hot path runs on this stream; the collective is ordered behind it
    torch.cuda.synchronize()
    step_no = [0]
    unreduced = [None]  # N > 1: the buffer of the batch queued last, not reduced yet

    # Who issues the collective.  native: the communicator of the C ABI (include/drprg_hip.h layout B) -- rank 0 makes the id,
    # the torch.distributed group that the launcher's environment set up carries its 128 bytes to the other ranks, every rank
    # calls drprg_hip_comm_init_rank, and drprg_hip_allreduce issues ONE ncclAllReduce(sum, u32) over the packed buffer
    # [coverage | reads per PRG] on a stream of its own (ordered against the hot path's stream by events, never by the host).
    # Under the one-GPU test hook (DRPRG_BENCH_BACKEND=gloo: every rank on GPU 0, which RCCL refuses as duplicate devices) every
    # rank's native communicator has ONE rank -- the call sequence runs, the sum over the ranks is gloo's.
    comm_mode = args.comm if args.comm != "auto" else ("native" if backend == "nccl" else "torch")
    native = None
    cstream = None
    comm_fallback = None
    if world > 1 and comm_mode == "native":
        from drprg_amd.distributed import NativeComm

        def exchange(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        native_error = None
        try:
            if backend == "nccl":
                native = NativeComm(rank, world, local_rank, exchange)
            else:
                native = NativeComm(0, 1, local_rank, lambda ident: ident)
        except Exception as e:  # (e.g. no RCCL library the binding can open)
            native_error = str(e)
        # every rank must take the same path: one that could not make its communicator sends all of them to torch.distributed
        flag = torch.tensor([0 if native is None else 1], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if native is not None:
                native.close()
                native = None
            comm_mode = "torch"
            comm_fallback = native_error or "another rank could not make its native communicator"
            if rank == 0:
                print("[bench] --comm native is not available (%s): reducing through torch.distributed" % comm_fallback, file=sys.stderr)
        else:
            cstream = torch.cuda.Stream(device)

    class _Done:  # what the stream of the hot path waits for before it reuses a buffer
        def __init__(self, event=None, work=None):
            self.event, self.work = event, work

        def wait(self):
            if self.work is not None:
                self.work.wait()  # (torch: the current stream waits, not the host)
            if self.event is not None:
                torch.cuda.current_stream().wait_event(self.event)

    def all_reduce_async(buf, after=None):
        """sum of `buf` over the ranks, asynchronous: returns what to wait for.  `after`: an event behind which the reduce
        must run (None: the buffer's batch is already complete on the device, as the host has seen)."""
        if native is None:
            if after is not None:
                torch.cuda.current_stream().wait_event(after)
            return _Done(work=dist.all_reduce(buf, async_op=True))
        if after is not None:
            cstream.wait_event(after)
        native.allreduce(ctx, buf.data_ptr(), buf.data_ptr() + 8 * ctx.n_knodes, cstream.cuda_stream)
        work = None
        if backend != "nccl":  # test hook: the one-rank native call ran; gloo sums over the ranks behind it
            ev = torch.cuda.Event()
            ev.record(cstream)
            ev.synchronize()
            work = dist.all_reduce(buf, async_op=True)
        ev = torch.cuda.Event()
        ev.record(cstream)
        return _Done(event=ev, work=work)

    def step(reduce=True):
        b = step_no[0] % len(accs)
        step_no[0] += 1
        acc = accs[b]
        with torch.cuda.stream(stream):
            if pending[b] is not None:
                pending[b].wait()  # (the stream waits, not the host)
                pending[b] = None
            acc.zero_()
            out = (acc.data_ptr(), acc.data_ptr() + 8 * ctx.n_knodes, stream.cuda_stream)
            if packed:
                ctx.map_device_packed(d_words.data_ptr(), offsets.data_ptr(), n_reads, n_bases, d_npos.data_ptr(), n_npos, *out, deferred=deferred)
            elif deferred:
                ctx.map_device_async(bases.data_ptr(), offsets.data_ptr(), n_reads, n_bases, *out)
            else:
                ctx.map_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_bases, *out)
            if world > 1 and reduce:
                if deferred:
                    if unreduced[0] is not None:  # the batch before this one is complete now
                        pending[unreduced[0]] = all_reduce_async(accs[unreduced[0]])
                    unreduced[0] = b
                else:
                    all_reduce_async(acc).wait()
        return acc

    def drain():
        ctx.sync()
        with torch.cuda.stream(stream):
            if unreduced[0] is not None:
                pending[unreduced[0]] = all_reduce_async(accs[unreduced[0]])
                unreduced[0] = None
            for b, w in enumerate(pending):
                if w is not None:
                    w.wait()
                    pending[b] = None

    def barrier():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    acc = accs[0]
    # ---- the device's state (round 6).  A MI355X that has idled while the host built the batch (seconds) takes hundreds of milliseconds of
    # work to reach the clocks it then holds: the same 20 steps take 0.49 ms each right after the preparation, 0.465 after 50 ms of mapping
    # and 0.445 after half a second (profiles/r06/spinup.txt) -- and W = 5 warm-up steps are 2.5 ms.  So the line carries both: "cold_start",
    # the W + K steps of this very command measured first, on the device as the preparation leaves it; and the headline, the same W + K steps
    # after --spinup-ms (600) of untimed mapping of the same batch -- what a process that maps sample after sample runs at. ----
    cold_start = None
    if world == 1 and not args.no_checks and args.spinup_ms > 0:
        for _ in range(args.warmup):
            step()
        drain()
        torch.cuda.synchronize()
        ctx.kernel_timing(enable=True, reset=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(args.steps):
            step()
        c1.record(stream)
        drain()
        torch.cuda.synchronize()
        c_ms, c_n = ctx.kernel_timing(enable=False)
        cold_ms = c0.elapsed_time(c1) / args.steps
        cold_start = {"ms_per_step": cold_ms, "value": n_reads / (cold_ms * 1e-3), "unit": "reads/s", "dominant_kernel_avg_launch_ms": c_ms / max(c_n, 1),
                      "how": f"{args.warmup} warm-up + {args.steps} timed steps (device time between two events) FIRST, before anything else ran on the "
                             "device since the batch was built: a device that has idled for seconds"}
    spinup_steps = 0
    if args.spinup_ms > 0:
        # (mapping only, no reduce: the loop runs by the clock, so the ranks of a job do not run it equally often -- and collectives are matched
        # by their order.  Until late in round 6 these steps reduced like the timed ones: eight ranks over gloo then waited for each other
        # until the transport's timeout, profiles/r06/README.md.)
        barrier()
        spin_ms = args.spinup_ms + rank * float(os.environ.get("DRPRG_BENCH_SPINUP_SKEW_MS", "0"))  # (test hook: ranks that surely differ in their step counts)
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
            for _ in range(8):
                step(reduce=False)
            spinup_steps += 8
            if not deferred:
                continue
            ctx.sync()  # (the host must not run far ahead of the device: the wall clock is the device's busy time)
        drain()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ctx.kernel_timing(enable=True, reset=True)
    # one event per step on the hot path's stream (recorded behind the step's last launch): the differences are the per-step
    # device times of the timed region (dispersion of the headline; the steps are queued back to back)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record(stream)
    # (an event record is a barrier packet on the stream: ~5 us of idle GPU per mark, 1 % of a 0.53 ms step when every step is marked --
    # profiles/r05/verify_scan.txt; so a mark per ten steps, each entry of step_ms then being the mean of its ten.  DRPRG_BENCH_MARK_EVERY=1: every step)
    # Round 6: five groups at least (a median of two was the larger one -- ADVICE r05): 20 steps = five groups of four
    mark_every = max(1, int(os.environ.get("DRPRG_BENCH_MARK_EVERY", str(max(1, args.steps // 5)) if args.steps >= 20 else "1")))
    marked = [0]
    for i in range(args.steps):
        acc = step()
        if (i + 1) % mark_every == 0 or i + 1 == args.steps:
            marks[i + 1].record(stream)
            marked.append(i + 1)
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[a].elapsed_time(marks[b]) / (b - a) for a, b in zip(marked, marked[1:]))
    covg, prg_reads = acc[: 2 * ctx.n_knodes], acc[2 * ctx.n_knodes:]  # the last step's (reduced) result
    k_ms, k_launches = ctx.kernel_timing(enable=False)
    filter_schedule = ctx.filter_schedule()  # (of the last timed step: how sketch_filter_kernel's tiles were handed out, when its wave classes ended)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    checksum = int(covg.to(torch.int64).sum().item())
    counters = ctx.counters()
    # N > 1: the vector every rank holds must be the sum of the ranks' own vectors (one more, untimed, pass + reduce)
    reduce_consistent = None
    if world > 1:
        own = torch.zeros_like(acc)
        with torch.cuda.stream(stream):
            ctx.map_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_bases, own.data_ptr(), own.data_ptr() + 8 * ctx.n_knodes,
                           stream.cuda_stream)
            # (through the OTHER binding than the timed steps used, when there are two: native and torch must agree)
            torch.cuda.synchronize()
            dist.all_reduce(own)
        torch.cuda.synchronize()
        ok = torch.tensor([1 if torch.equal(own, acc) else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        reduce_consistent = bool(ok.item())
    shard_invariant = kernels_agree = packed_equals_ascii = None
    if world == 1 and not args.no_checks:
        if packed:  # the packed batch must give what the ASCII batch gives, at full size
            ref = torch.zeros_like(acc)
            map_range(ctx, bases, offsets, 0, n_reads, ref[: 2 * ctx.n_knodes], ref[2 * ctx.n_knodes:], stream, torch)
            packed_equals_ascii = bool(torch.equal(ref, acc))
        shard_invariant, kernels_agree = full_size_checks(torch, ctx, opts, bases, offsets, n_reads, covg, prg_reads, stream)
    # The same batch in the OTHER input format, outside the timed region: the headline is the ASCII batch (B(L) = L + 8 bytes per read are
    # really read); the executables' ingest hands over 2-bit packed words, so the line carries that step as well ("packed_input").
    packed_leg = None
    if world == 1 and not args.no_checks and not packed and os.environ.get("DRPRG_BENCH_PACKED_LEG", "1") != "0":
        pw = torch.zeros((n_bases + 15) // 16 + 4, dtype=torch.int32, device=device)
        pn = torch.zeros(1 << 16, dtype=torch.int64, device=device)
        torch.cuda.synchronize()  # (as above)
        pn_n = ctx.pack_device(bases.data_ptr(), n_bases, pw.data_ptr(), pn.data_ptr(), pn.numel())
        bufs = [torch.zeros_like(acc) for _ in range(2)]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        warm = max(args.warmup, 1)
        with torch.cuda.stream(stream):
            def pstep(i):
                b = bufs[i % 2]
                b.zero_()
                ctx.map_device_packed(pw.data_ptr(), offsets.data_ptr(), n_reads, n_bases, pn.data_ptr(), pn_n, b.data_ptr(),
                                      b.data_ptr() + 8 * ctx.n_knodes, stream.cuda_stream, deferred=deferred)
            # (the checks before this leg ran other kernels on smaller batches: the leg gets a short spin-up of its own -- without it the line's
            # packed step read 0.345-0.349 ms where `--input packed`, behind the full spin-up, reads 0.326-0.330)
            t_sp = time.perf_counter()
            while (time.perf_counter() - t_sp) * 1e3 < min(args.spinup_ms, 200.0):
                for q in range(8):
                    pstep(q)
                ctx.sync()
            for i in range(-warm, args.steps):
                pstep(i)
                if i == -1:
                    ctx.sync()
                    torch.cuda.synchronize()
                    ctx.kernel_timing(enable=True, reset=True)
                    evs[0].record(stream)
                    p_marked = [0]
                elif i >= 0 and ((i + 1) % mark_every == 0 or i + 1 == args.steps):  # (marks as in the headline's region)
                    evs[i + 1].record(stream)
                    p_marked.append(i + 1)
        ctx.sync()
        torch.cuda.synchronize()
        pk_ms, pk_n = ctx.kernel_timing(enable=False)
        pk_schedule = ctx.filter_schedule()
        p_steps = sorted(evs[a].elapsed_time(evs[b]) / (b - a) for a, b in zip(p_marked, p_marked[1:]))
        p_ms = evs[0].elapsed_time(evs[args.steps]) / args.steps
        packed_leg = {"ms_per_step": p_ms, "value": n_reads / (p_ms * 1e-3), "unit": "reads/s",
                      "step_ms": {"min": p_steps[0], "median": p_steps[len(p_steps) // 2], "max": p_steps[-1]},
                      "dominant_kernel_avg_launch_ms": pk_ms / max(pk_n, 1), "filter_schedule": pk_schedule,
                      "coverage_equals_the_ascii_run": bool(torch.equal(bufs[(args.steps - 1) % 2], acc)),
                      "bytes_of_the_batch_as_stored": (n_bases + 15) // 16 * 4 + 8 * (n_reads + 1),
                      "how": "the batch packed on the device before this leg (drprg_hip_pack_device), then the same number of steps through "
                             "drprg_hip_map_device_packed, device time between HIP events on the hot path's stream; outside the timed region of the headline"}
        del pw, pn, bufs

    # Different batches in rotation, outside the timed region: the headline maps ONE resident batch again and again, and a batch that comes round
    # again finds a little of itself in the memory-side cache (256 MB).  What a stream of different batches runs at ("other_batches_in_rotation";
    # profiles/r06/schedule.txt section 11: 1.3-2.8 % below the repeated batch for ASCII input, nothing for packed).
    rotation_leg = None
    if world == 1 and not args.no_checks and not packed and os.environ.get("DRPRG_BENCH_ROTATION_LEG", "1") != "0":
        hp = torch.from_numpy(genomes.padded()).to(device)
        rot = [(bases, offsets, n_bases)]
        for sd in (101, 102):
            if args.workload == "nanopore":
                b2, o2 = gpu_sample_long_reads(torch, hp, hap_lens, n_reads, sd, device)
            else:
                b2, o2 = gpu_sample_reads(torch, hp, hap_lens, n_reads, args.read_len, sd, device)
            rot.append((b2, o2, int(b2.numel())))
        del hp
        torch.cuda.synchronize()
        bufs = [torch.zeros_like(acc) for _ in range(2)]
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        warm = max(args.warmup, 3)
        with torch.cuda.stream(stream):
            for i in range(-warm, args.steps):
                b = bufs[i % 2]
                b.zero_()
                rb, ro, rn = rot[i % len(rot)]
                (ctx.map_device_async if deferred else ctx.map_device)(rb.data_ptr(), ro.data_ptr(), n_reads, rn, b.data_ptr(), b.data_ptr() + 8 * ctx.n_knodes,
                                                                       stream.cuda_stream)
                if i == -1:
                    ctx.sync()
                    torch.cuda.synchronize()
                    ctx.kernel_timing(enable=True, reset=True)
                    r0.record(stream)
            r1.record(stream)
        ctx.sync()
        torch.cuda.synchronize()
        rk_ms, rk_n = ctx.kernel_timing(enable=False)
        r_ms = r0.elapsed_time(r1) / args.steps
        rotation_leg = {"ms_per_step": r_ms, "value": n_reads / (r_ms * 1e-3), "unit": "reads/s", "batches": len(rot),
                        "dominant_kernel_avg_launch_ms": rk_ms / max(rk_n, 1),
                        "how": f"{len(rot)} different batches of the workload (other seeds) mapped in rotation, the same number of steps, device time between "
                               "HIP events on the hot path's stream; outside the timed region of the headline, which repeats one resident batch"}
        del rot, bufs

    if rank == 0:
        total_reads = n_reads * world * args.steps
        value = total_reads / elapsed
        # algorithmic bytes per launch: sum(L + 8) over the reads + index table + 8 B per k-mer node
        table_bytes = ctx.n_slots * (4 + 8) + ctx.n_records * (4 + 2)
        alg_bytes = n_bases + 8 * n_reads + table_bytes + 8 * ctx.n_knodes
        # a batch may be mapped as several read ranges (one launch of the dominant kernel each, on concurrent streams): the
        # roofline prices one launch = its share of the batch's algorithmic bytes over its own duration (HIP events)
        launches_per_step = max(k_launches, 1) / args.steps
        avg_ms = k_ms / max(k_launches, 1)
        alg_bytes = alg_bytes / launches_per_step
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kernel_name = {1: "sketch_probe_kernel", 2: "sketch_filter_kernel", 3: "sketch_probe_kernel"}.get(counters.get("kernel"), "?")
        if counters.get("kernel") == 3 and K == 15 and W in (11, 14) and os.environ.get("DRPRG_DIRECT_FORM") != "lds":
            kernel_name = "sketch_wave_kernel"  # the register-resident form of the direct kernel (csrc/sketch_wave.hip)
        # HBM bytes per launch of that kernel from rocprofv3 PMC counters (separate --pmc passes, FETCH_SIZE doubled as
        # the microarch guide prescribes for wide coalesced loads on gfx950): measured offline, committed under profiles/
        # (the replayed numbers belong to the kernel source they were measured on: an entry whose sources have changed since -- or that
        # does not say what it was measured on -- is dropped, with the reason; tools/make_profile_json.py writes the entries)
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        wkey = args.workload + ("-packed" if packed else "")
        if n_reads == default_reads and os.path.exists(tfile):
            entry = json.load(open(tfile)).get(wkey, {}).get(kernel_name, {})
            stale = profile_entry_stale(entry, kernel_name)
            if entry.get("hbm_bytes_per_launch") is not None and not stale:
                traffic = entry["hbm_bytes_per_launch"] / launches_per_step  # (measured per batch)
                traffic_source = (f"profiles/traffic.json, rows {entry.get('rows')}, kernel sources sha256 {entry.get('source_sha16')} = the build's "
                                  "(offline rocprofv3 PMC passes of this workload, FETCH_SIZE doubled + WRITE_SIZE)")
            elif stale:
                traffic_source = "dropped: " + stale
        # secondary bound (SURVEY 8d "report honestly"): the dominant kernel's VALU issue slots.  Wave instructions per launch come
        # from offline rocprofv3 SQ-counter passes of this workload (profiles/valu.json, SQ_INSTS_VALU); the duration is the live one.
        # One wave64 VALU instruction of the kinds the sketch kernels are made of occupies its SIMD for 4 cycles, however many waves
        # share the SIMD (measured at 1 / 2 / 4 / 8 waves per SIMD: tools/mb_issue.hip -> profiles/r04/mb_issue.txt: v_lshl_add_u32 4.2,
        # v_bfe_u32 4.4, v_bitop3_b32 4.0, v_min3_u32 4.5, v_dot4_u32_u8 4.1, DPP moves 4.5; plain v_add / v_xor / v_and / v_not reach 2.5 in
        # a stream of their own and 4.2 interleaved with VOP3 instructions); 1024 SIMDs at 2.4 GHz.
        secondary = None
        vfile = os.path.join(ROOT, "profiles", "valu.json")
        if n_reads == default_reads and os.path.exists(vfile) and avg_ms > 0:
            entry = json.load(open(vfile)).get(wkey, {}).get(kernel_name, {})
            v = entry.get("valu_wave_insts_per_launch")
            stale = profile_entry_stale(entry, kernel_name)
            if v and stale:
                secondary = {"bound": "valu_issue", "wave_insts": None, "frac": None, "source": "dropped: " + stale}
            elif v:
                v = v / launches_per_step
                peak = 1024 * 2.4e9 / 4
                secondary = {"bound": "valu_issue", "wave_insts": v, "achieved": v / (avg_ms * 1e-3), "peak": peak, "unit": "wave-instructions/s",
                             "frac": v / (avg_ms * 1e-3) / peak,
                             "source": f"wave_insts: profiles/valu.json, rows {entry.get('rows')}, kernel sources sha256 {entry.get('source_sha16')} = the "
                                       "build's (offline rocprofv3 --pmc SQ_INSTS_VALU of this workload) over the live kernel "
                                       "duration; peak: 4 cycles per wave64 integer VALU instruction per SIMD, measured at 1/2/4/8 waves per SIMD in "
                                       "profiles/r04/mb_issue.txt (tools/mb_issue.hip)"}
        out = {
            "metric": "reads/sec (+ achieved HBM GB/s) predicting on mtb index, 1/2/4/8 GPUs",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "timed_region_s": elapsed,
            "step_ms": {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1],
                        "groups": len(step_ms), "how": "HIP events on the hot path's stream behind the last launch of every %d%s step (device time per step%s)" % (mark_every, {1: "st", 2: "nd", 3: "rd"}.get(mark_every, "th"), "" if mark_every == 1 else ", mean of each group")},
            "spinup": {"ms": args.spinup_ms, "steps": spinup_steps,
                       "what": "untimed steps on the same batch before the warm-up steps: the device at the clocks it holds under load (cold_start: without)"},
            "cold_start": cold_start,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{cfg_name}: {cfg_desc}",
                "input_format": ("packed: 2 bits per base, u32 words + u64 read offsets + sparse non-ACGT positions (HBM-resident; "
                                 f"{(n_bases + 15) // 16 * 4 + 8 * (n_reads + 1)} bytes per batch)" if packed
                                 else "ascii: one byte per base + u64 read offsets"),
                "full_size_packed_equals_ascii": packed_equals_ascii,
                "reads_per_gpu": n_reads, "bases_per_gpu": n_bases, "mean_read_len": n_bases / max(n_reads, 1), "w": W, "k": K,
                "loci": ctx.n_prgs, "index_keys": ctx.n_keys, "kmer_nodes": ctx.n_knodes, "sharding": f"reads x{world}",
                "collective": (("one all_reduce(sum, u32) of [coverage | reads per PRG] per step, overlapped with the next step's mapping" if overlap
                                else "one all_reduce(sum, u32) of [coverage | reads per PRG] per step") if world > 1 else "none"),
                "comm": (None if world == 1 else
                         ("native: drprg_hip_comm_unique_id / comm_init_rank / drprg_hip_allreduce (C ABI, RCCL bound at run time), id broadcast "
                          "over the torch.distributed group" + ("" if backend == "nccl" else "; ONE-GPU TEST HOOK: one-rank native communicators, "
                                                                 "the sum over the ranks by gloo"))
                         if native is not None else f"torch.distributed.all_reduce ({backend})"
                         + (f" -- the native communicator could not be made: {comm_fallback}" if comm_fallback else "")),
                "reduced_words": n_acc,
                "all_ranks_hold_the_sum_of_the_ranks_vectors": reduce_consistent,
                "bases_per_s": n_bases * world * args.steps / elapsed,
                "hits_per_batch": counters.get("hits", 0) // max(step_no[0], 1),  # (every step() so far: cold leg, spin-up, warm-up, timed)
                "clusters_kept_per_batch": counters.get("clusters_kept", 0) // max(step_no[0], 1),
                "leftover_reads_per_batch": counters.get("leftover_reads", 0) // max(step_no[0], 1),
                "filter_tiers": ctx.table_tier(),
                "coverage_checksum": checksum, "full_size_shard_invariance": shard_invariant,
                "full_size_direct_vs_filtered_kernel_identical": kernels_agree,
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                "achievable": HBM_ACHIEVABLE_GBS, "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_of_the_batch_as_stored": ((n_bases + 15) // 16 * 4 if packed else n_bases) + 8 * (n_reads + 1),
                "avg_launch_ms": avg_ms, "launches_timed": k_launches, "launches_per_step": launches_per_step,
                "secondary": secondary,
                "filter_schedule": filter_schedule,
            },
        }
        if packed_leg is not None:
            packed_leg["roofline_frac_priced_on_L_plus_8"] = (alg_bytes / (packed_leg["dominant_kernel_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                              if packed_leg["dominant_kernel_avg_launch_ms"] > 0 else None)
            # ... and on the bytes the packed kernel really reads (the batch as stored + index table + coverage vector): SURVEY 8d allows the
            # L + 8 pricing ("packing is an optimisation, not a change of work"); this is the fraction of the memory system it uses
            packed_read = packed_leg["bytes_of_the_batch_as_stored"] + table_bytes + 8 * ctx.n_knodes
            packed_leg["bytes_read_per_launch"] = packed_read
            packed_leg["frac_of_bytes_read"] = (packed_read / (packed_leg["dominant_kernel_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                if packed_leg["dominant_kernel_avg_launch_ms"] > 0 else None)
            out["packed_input"] = packed_leg
        if rotation_leg is not None:
            out["other_batches_in_rotation"] = rotation_leg
        # CPU baseline: the oracle (a scalar port of the same path, oracle/oracle.c) on a bounded sample of rank 0's shard,
        # once on one thread and once with the reads split over the host's cores (threads calling the same C function on
        # disjoint read ranges; integer coverage sums commute).  The multi-thread sample is as large as ~10 s allow -- with
        # enough cores the whole batch -- and the HIP path must give the identical vector on it.
        if args.cpu_sample != 0 and world == 1:  # (rank 0 at N=1 only)
            from concurrent.futures import ThreadPoolExecutor
            from util import Oracle, cluster_fraction, map_params
            orc = Oracle()
            idx = orc.build_index(panel.prgs, W, K)  # the oracle's own index of the same PRG strings
            md, er = map_params(K, illumina)
            frac = cluster_fraction(er, K)
            mean_len = max(n_bases / n_reads, 1.0)
            ns1 = args.cpu_sample if args.cpu_sample > 0 else max(1, int(2_000_000 * 150 / mean_len))
            ns1 = min(ns1, n_reads)
            threads = max(1, min(os.cpu_count() or 1, 64))
            ns = min(n_reads, ns1 * threads if args.cpu_sample < 0 else ns1, max(ns1, int(3e9 / mean_len)))
            nb = int(offsets[ns].item())
            hb = bases[:nb].cpu().numpy()
            ho = offsets[:ns + 1].cpu().numpy().astype(np.uint64)

            def cpu_map(lo, hi):
                return orc.map_reads(hb[int(ho[lo]):int(ho[hi])], ho[lo:hi + 1] - ho[lo], idx, W, K, md, frac, 10)[:2]

            t1 = time.perf_counter()
            cpu_map(0, ns1)
            cpu1_s = time.perf_counter() - t1
            cuts = [ns * i // threads for i in range(threads + 1)]
            t1 = time.perf_counter()
            with ThreadPoolExecutor(threads) as pool:
                parts = list(pool.map(lambda i: cpu_map(cuts[i], cuts[i + 1]), range(threads)))
            cpu_s = time.perf_counter() - t1
            ocov = np.sum(np.stack([p[0] for p in parts]).astype(np.uint64), axis=0).astype(np.uint32)
            oprg = np.sum(np.stack([p[1] for p in parts]).astype(np.uint64), axis=0).astype(np.uint32)
            # the same sample through the HIP path must give the identical vector
            c2 = torch.zeros_like(covg)
            p2 = torch.zeros_like(prg_reads)
            torch.cuda.synchronize()
            map_range(ctx, bases, offsets, 0, ns, c2, p2, stream, torch)
            parity = bool(np.array_equal(c2.cpu().numpy().view(np.uint32), ocov))
            # ... and the genotyped VCF of that vector (SURVEY 8d "parity gates run with every benchmark"): the product's genotyper on the
            # coverage the DEVICE accumulated against the oracle's own model / site enumeration / statistics / likelihoods on the
            # oracle's vector (tests/util.py oracle_vcf_text), byte for byte minus ##fileDate
            vcf_parity = None
            try:
                from util import oracle_vcf_text, vcf_without_date
                host = Context(prg, W, K, device=-1, from_files=False)
                host.set_opts(illumina=illumina, genome_size=opts["genome_size"])
                host.set_coverage(c2.cpu().numpy().view(np.uint32), p2.cpu().numpy().view(np.uint32), nb)
                vcf_path = os.path.join(tmp, "pandora_genotyped.vcf")
                host.genotype(os.path.join(tmp, "genes.fa"), vcf_path)
                want, _ = oracle_vcf_text(orc, panel.names, panel.prgs, dict(zip(panel.names, panel.refs)), ocov, oprg, nb, W, K, opts["genome_size"], er)
                vcf_parity = bool(vcf_without_date(vcf_path) == want)
                host.close()
            except Exception as exc:  # (the bench line must not depend on this gate)
                vcf_parity = f"not run: {type(exc).__name__}: {exc}"
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "reads/s", "cores": threads, "kind": "port",
                                   "sample": f"first {ns} reads ({nb} bases) of rank 0's shard, oracle/oracle.c on {threads} threads "
                                             f"(disjoint read ranges), {cpu_s:.1f}s",
                                   "single_thread_value": ns1 / cpu1_s, "single_thread_sample": f"first {ns1} reads, {cpu1_s:.1f}s",
                                   "host_cores_available": os.cpu_count(), "cpu_model": cpu_model(), "parity_vs_hip_on_sample": parity,
                                   "genotyped_vcf_identical_to_oracle_on_sample": vcf_parity}
        if args.e2e and world == 1 and args.workload != "nanopore" and not args.no_checks:
            try:
                out["e2e"] = e2e_leg(torch, ctx, synth, bases, n_reads, args.read_len, covg)
            except Exception as exc:  # (no room in /dev/shm, ...: the bench line must not depend on this leg)
                out["e2e"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if native is not None:
        native.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
