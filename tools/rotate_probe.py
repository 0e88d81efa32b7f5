#!/usr/bin/env python3
"""Does mapping the SAME batch again and again (what bench.py's steps do) flatter the kernel?  R different 10 M-read batches mapped in rotation
against one batch mapped repeatedly, alternating, same process: ms per step (HIP events) and the dominant kernel's live average.

usage: python tools/rotate_probe.py [R=4] [steps=200] [ascii|packed]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from drprg_amd import Context, synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
packed = len(sys.argv) > 3 and sys.argv[3] == "packed"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
panel = bench.make_panel(synth, bench.WORKLOADS["mtb"][4])
tmp = tempfile.mkdtemp(prefix="drprg_rot_")
prg = os.path.join(tmp, "dr.prg")
panel.write(prg, os.path.join(tmp, "genes.fa"))
ctx = Context(prg, 11, 15, device=0, from_files=False, threads=8)
ctx.set_opts(kernel=0, illumina=True, min_cluster_size=10, genome_size=synth.MTB_GENOME_SIZE)
genomes = synth.HaplotypeGenomes(panel, n_hap=8)
hap_pad = torch.from_numpy(genomes.padded()).to(dev)
hap_lens = torch.from_numpy(genomes.lens).to(dev)
n_reads = 10_000_000
batches = []
for r in range(R):
    b, o = bench.gpu_sample_reads(torch, hap_pad, hap_lens, n_reads, 150, 2 + r, dev)
    w = npos = None
    n_npos = 0
    torch.cuda.synchronize()  # (the context's calls run on ITS stream: what torch queued must be through)
    if packed:
        w = torch.zeros((b.numel() + 15) // 16 + 4, dtype=torch.int32, device=dev)
        npos = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        n_npos = ctx.pack_device(b.data_ptr(), b.numel(), w.data_ptr(), npos.data_ptr(), npos.numel())
    batches.append((b, o, w, npos, n_npos))
    torch.cuda.synchronize()
    print("batch", r, b.numel(), o.numel(), int(o[-1].item()), file=sys.stderr, flush=True)
n_acc = 2 * ctx.n_knodes + ctx.n_prgs
accs = [torch.zeros(n_acc, dtype=torch.int32, device=dev) for _ in range(2)]
stream = torch.cuda.Stream(dev)
no = [0]


def step(i):
    b, o, w, npos, n_npos = batches[i]
    acc = accs[no[0] % 2]
    no[0] += 1
    with torch.cuda.stream(stream):
        acc.zero_()
        out = (acc.data_ptr(), acc.data_ptr() + 8 * ctx.n_knodes, stream.cuda_stream)
        if packed:
            ctx.map_device_packed(w.data_ptr(), o.data_ptr(), n_reads, b.numel(), npos.data_ptr(), n_npos, *out, deferred=True)
        else:
            ctx.map_device_async(b.data_ptr(), o.data_ptr(), n_reads, b.numel(), *out)


def run(rotate, steps):
    for i in range(16):
        step(i % R if rotate else 0)
    ctx.sync()
    ctx.kernel_timing(enable=True, reset=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(steps):
        step(i % R if rotate else 0)
        if i % 8 == 7:
            ctx.sync()
    e1.record(stream)
    ctx.sync()
    torch.cuda.synchronize()
    ms, n = ctx.kernel_timing(enable=False)
    return e0.elapsed_time(e1) / steps, ms / max(n, 1)


for i in range(600):  # the device to its clocks
    step(0)
    if i % 8 == 7:
        ctx.sync()
        if os.environ.get("ROT_DEBUG"):
            torch.cuda.synchronize()
            print("spin", i, file=sys.stderr, flush=True)
ctx.sync()
for rnd in range(4):
    for rotate in (False, True):
        s, k = run(rotate, STEPS)
        print(f"round {rnd} {'rotating over %d batches' % R if rotate else 'one batch repeated    '}: {s:.4f} ms/step, dominant kernel {k:.4f} ms", flush=True)
