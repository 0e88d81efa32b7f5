#!/usr/bin/env python3
"""End-to-end timing of the drop-in `pandora map` executable on a synthetic FASTQ (plain and .gz) on the GPU box.
Usage: python tools/e2e_cli_timing.py [n_reads] [threads]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from drprg_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
tmp = tempfile.mkdtemp(prefix="drprg_e2e_")
panel = synth.mtb_like_panel()
prg, genes = os.path.join(tmp, "dr.prg"), os.path.join(tmp, "genes.fa")
panel.write(prg, genes)
gen = synth.HaplotypeGenomes(panel, n_hap=8)
t = time.time()
bases, offs = synth.sample_short_reads(gen, n, seed=2)
fq = os.path.join(tmp, "reads.fq")
synth.write_fastq_fixed(fq, bases, 150)
print(f"generated {n} reads, {os.path.getsize(fq) / 1e6:.0f} MB FASTQ in {time.time() - t:.1f}s", flush=True)
exe = os.path.join(ROOT, "drprg_amd", "bin", "pandora")
subprocess.run([exe, "index", "-t", "8", "-w", "11", "-k", "15", prg], check=True, stdout=subprocess.DEVNULL)
def run(reads, label):
    out = os.path.join(tmp, "out_" + label)
    t = time.time()
    r = subprocess.run([exe, "map", "--genotype", "--local", "--gt-conf", "0", "-o", out, "-g", "4411532", "--max-covg", "4294967295",
                        "--vcf-refs", genes, "-t", threads, "-w", "11", "-k", "15", "-c", "10", "-I", prg, reads], capture_output=True, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if "reads=" in l]
    print(f"{label}: wall {dt:.2f}s -> {n / dt / 1e6:.2f} M reads/s end to end (process start to VCF) | {line[0] if line else ''}", flush=True)
    return open(os.path.join(out, "pandora_genotyped.vcf")).read().split("\n", 2)[2]
a = run(fq, "plain")
if n <= 4_000_000:
    subprocess.run(["gzip", "-1", "-k", fq], check=True)
    b = run(fq + ".gz", "gzip")
    assert a == b
