#!/usr/bin/env python3
"""End-to-end leg alone with the ingest's own timers (DRPRG_INGEST_DEBUG=1): 10 M x 150 bp FASTQ text in /dev/shm -> coverage."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drprg_amd import Context, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
panel = synth.mtb_8d_panel()
d = tempfile.mkdtemp(prefix="e2e_", dir="/dev/shm")
try:
    prg = os.path.join(d, "dr.prg"); panel.write(prg)
    ctx = Context(prg, 11, 15, device=0, from_files=False, threads=8)
    ctx.set_opts(illumina=True, genome_size=synth.MTB_GENOME_SIZE)
    gen = synth.HaplotypeGenomes(panel, n_hap=4)
    bases, offs = synth.sample_short_reads(gen, n, seed=2)
    fq = os.path.join(d, "r.fq"); synth.write_fastq_fixed(fq, bases, 150)
    ctx.set_threads(threads)
    # both ingest formats in turn (E2E_FORMATS=ascii,packed), alternating, so that the host's load hits both alike
    formats = os.environ.get("E2E_FORMATS", "ascii,packed").split(",")
    best = {f: 1e9 for f in formats}
    for rep in range(int(os.environ.get("E2E_REPS", "4"))):
        for f in formats:
            ctx.set_input_format(f == "packed")
            ctx.reset()
            t = time.perf_counter(); ctx.map_fastx(fq); dt = time.perf_counter() - t
            best[f] = min(best[f], dt)
            print("rep %d %-6s: %.1f ms  %.2e reads/s" % (rep, f, dt * 1e3, n / dt), flush=True)
    print("best:", {f: "%.1f ms" % (v * 1e3) for f, v in best.items()})
finally:
    shutil.rmtree(d, ignore_errors=True)
