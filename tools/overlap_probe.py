"""does running two batches at once pay on this workload?  Two contexts on two streams, one host thread each, against one
context alone (run under gpurun): python tools/overlap_probe.py [big|mtb|nanopore]"""
import os, sys, time, threading, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from drprg_amd import Context, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "big"
_, _, n_reads, illumina, panel_name = bench.WORKLOADS[wl]
panel = {"mtb_8d": synth.mtb_8d_panel, "mtb_like": synth.mtb_like_panel, "big": synth.big_panel}[panel_name]()
tmp = tempfile.mkdtemp(); prg = os.path.join(tmp, "dr.prg"); panel.write(prg)
dev = torch.device("cuda", 0)
genomes = synth.HaplotypeGenomes(panel, n_hap=8)
hap_pad = torch.from_numpy(genomes.padded()).to(dev); hap_lens = torch.from_numpy(genomes.lens).to(dev)
W, K = (11, 15) if illumina else (14, 15)
sets = []
for i in range(2):
    ctx = Context(prg, W, K, device=0, from_files=False, threads=8)
    ctx.set_opts(illumina=illumina, min_cluster_size=10, genome_size=synth.MTB_GENOME_SIZE)
    if wl == "nanopore":
        b, o = bench.gpu_sample_long_reads(torch, hap_pad, hap_lens, n_reads, 3 + i, dev)
    else:
        b, o = bench.gpu_sample_reads(torch, hap_pad, hap_lens, n_reads, 150, 2 + i, dev)
    acc = torch.zeros(2 * ctx.n_knodes + ctx.n_prgs, dtype=torch.int32, device=dev)
    sets.append((ctx, b, o, acc, torch.cuda.Stream(dev)))
torch.cuda.synchronize()
def loop(s, steps):
    ctx, b, o, acc, st = s
    for _ in range(steps):
        ctx.map_device(b.data_ptr(), o.data_ptr(), n_reads, int(b.numel()), acc.data_ptr(), acc.data_ptr() + 8 * ctx.n_knodes, st.cuda_stream)
for s in sets: loop(s, 2)
torch.cuda.synchronize()
steps = 8
t = time.perf_counter(); loop(sets[0], steps); torch.cuda.synchronize(); one = (time.perf_counter() - t) / steps
t = time.perf_counter()
th = [threading.Thread(target=loop, args=(s, steps)) for s in sets]
[x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
two = (time.perf_counter() - t) / (2 * steps)
print(f"{wl}: one context {one * 1e3:.3f} ms per batch; two contexts at once {two * 1e3:.3f} ms per batch ({one / two:.2f}x)")
