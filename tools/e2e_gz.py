"""end-to-end gzip timing on the GPU box: 4M reads as plain text / BGZF / one plain gzip stream through map_fastx (run under gpurun)"""
import os, sys, time, tempfile, gzip, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from drprg_amd import Context, synth
panel = synth.mtb_8d_panel()
tmp = tempfile.mkdtemp(dir="/dev/shm"); prg = os.path.join(tmp, "dr.prg"); panel.write(prg)
gen = synth.HaplotypeGenomes(panel, n_hap=4)
n = 4_000_000
bases, offs = synth.sample_short_reads(gen, n, seed=2)
fq = os.path.join(tmp, "r.fq"); synth.write_fastq_fixed(fq, bases, 150)
text = open(fq, "rb").read()
bg = os.path.join(tmp, "r.bgzf.gz"); bench.write_bgzf(bg, text)
import subprocess
gz = os.path.join(tmp, "r.plain.gz")
with open(gz, "wb") as fh:
    subprocess.run(["gzip", "-6", "-c", fq], stdout=fh, check=True)
print("sizes", os.path.getsize(fq), os.path.getsize(bg), os.path.getsize(gz), flush=True)
for threads in [int(x) for x in os.environ.get("E2E_THREADS", "8,16,32,64").split(",")]:
    ctx = Context(prg, 11, 15, device=0, from_files=False, threads=8)
    ctx.set_opts(illumina=True); ctx.set_threads(threads)
    ctx.map_fastx(fq)
    for path in (fq, bg, gz):
        ctx.reset(); t = time.perf_counter(); ctx.map_fastx(path); dt = time.perf_counter() - t
        print(threads, os.path.basename(path), "%.3fs %.2e reads/s" % (dt, n / dt), flush=True)
    ctx.close()
shutil.rmtree(tmp)
