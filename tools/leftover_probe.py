import os, sys, tempfile
sys.path.insert(0, '/root/repo') if os.path.isdir('/root/repo') else None
sys.path.insert(0, os.getcwd())
import torch, bench
from drprg_amd import Context, synth
for wl in ("big", "nanopore", "mtb"):
    _, _, n_reads, illumina, panel_name = bench.WORKLOADS[wl]
    n_reads //= 4
    panel = {"mtb_8d": synth.mtb_8d_panel, "big": synth.big_panel}[panel_name]()
    tmp = tempfile.mkdtemp(); prg = os.path.join(tmp, "dr.prg"); panel.write(prg)
    dev = torch.device("cuda", 0)
    genomes = synth.HaplotypeGenomes(panel, n_hap=8)
    hap_pad = torch.from_numpy(genomes.padded()).to(dev); hap_lens = torch.from_numpy(genomes.lens).to(dev)
    ctx = Context(prg, 11, 15, device=0, from_files=False, threads=8)
    ctx.set_opts(illumina=illumina, min_cluster_size=10, genome_size=synth.MTB_GENOME_SIZE)
    if wl == "nanopore": b, o = bench.gpu_sample_long_reads(torch, hap_pad, hap_lens, n_reads, 3, dev)
    else: b, o = bench.gpu_sample_reads(torch, hap_pad, hap_lens, n_reads, 150, 2, dev)
    torch.cuda.synchronize()
    ctx.map_device(b.data_ptr(), o.data_ptr(), n_reads, int(b.numel()))
    c = ctx.counters()
    print(wl, n_reads, {k: c[k] for k in ("hits", "clusters_kept", "leftover_reads", "kernel")})
