"""diagnostic: counters of the 500-locus workload with and without the in-kernel clustering (run under gpurun)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from drprg_amd import Context, synth
panel = synth.big_panel()
tmp = tempfile.mkdtemp(); prg = os.path.join(tmp, "dr.prg"); panel.write(prg)
genomes = synth.HaplotypeGenomes(panel, n_hap=8)
dev = torch.device("cuda", 0)
hp = torch.from_numpy(genomes.padded()).to(dev); hl = torch.from_numpy(genomes.lens).to(dev)
n = 2_000_000
bases, offs = bench.gpu_sample_reads(torch, hp, hl, n, 150, 2, dev)
for fuse in ("0", "1", "2"):
    os.environ["DRPRG_WAVE_FUSE"] = fuse
    ctx = Context(prg, 11, 15, device=0, from_files=False, threads=8)
    ctx.set_opts(illumina=True, genome_size=synth.MTB_GENOME_SIZE)
    ctx.map_device(bases.data_ptr(), offs.data_ptr(), n, int(bases.numel()))
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.map_device(bases.data_ptr(), offs.data_ptr(), n, int(bases.numel()))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    c = ctx.counters(); cov = ctx.coverage()[0]
    print("fuse", fuse, "ms %.2f" % (dt * 1e3), {k: c[k] for k in ("hits", "clusters_kept", "hits_kept", "leftover_reads")}, int(cov.astype(np.int64).sum()))
    ctx.close()
