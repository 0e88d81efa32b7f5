import os, sys, tempfile, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drprg_amd import Context, synth
import bench
panel = synth.mtb_like_panel(); tmp = tempfile.mkdtemp(); prg = tmp + "/dr.prg"; panel.write(prg)
ctx = Context(prg, 11, 15, device=0, from_files=False, threads=8); ctx.set_opts(illumina=True)
gen = synth.HaplotypeGenomes(panel, n_hap=8)
dev = torch.device("cuda", 0)
bases, offs = bench.gpu_sample_reads(torch, torch.from_numpy(gen.padded()).to(dev), torch.from_numpy(gen.lens).to(dev), 10_000_000, 150, 2, dev)
torch.cuda.synchronize()
ctx.map_device(bases.data_ptr(), offs.data_ptr(), 10_000_000, bases.numel())
c = ctx.counters(); m = c["minimizers"]
print(c, "cands", m & ((1 << 40) - 1), "raw", m >> 40, "tiles", bases.numel() / 8160, "keys", ctx.n_keys)
