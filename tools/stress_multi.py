"""stress of the multi-device ingest (device 0 listed three times): the same file mapped N times, every coverage vector compared with the
single-device one.  Usage: python tools/stress_multi.py [n_rounds] [n_reads]"""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drprg_amd import Context, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_500_000
tmp = tempfile.mkdtemp(dir="/dev/shm")
panel = synth.small_panel(seed=5, n_loci=6, length=700, site_every=60)
prg, genes = os.path.join(tmp, "dr.prg"), os.path.join(tmp, "genes.fa")
panel.write(prg, genes)
gen = synth.HaplotypeGenomes(panel, genome_size=60000, n_hap=4, seed=3)
bases, offs = synth.sample_short_reads(gen, n, seed=4)
fq = os.path.join(tmp, "reads.fq")
synth.write_fastq_fixed(fq, bases, 150)
single = Context(prg, 11, 15, device=0, from_files=False)
single.set_opts(illumina=True, genome_size=60000)
single.set_threads(8)
single.map_fastx(fq)
want, want_prg = single.coverage()
bad = 0
for devs in ([0, 0, 0], [0, 0], [0]):
    multi = Context(prg, 11, 15, from_files=False, devices=devs) if len(devs) > 1 else Context(prg, 11, 15, device=0, from_files=False)
    multi.set_opts(illumina=True, genome_size=60000)
    multi.set_threads(int(os.environ.get("STRESS_THREADS", "8")))
    for r in range(rounds):
        if os.environ.get("STRESS_COLD"):  # a fresh context every round: first-call allocations and staging growth every time
            multi.close()
            multi = Context(prg, 11, 15, from_files=False, devices=devs) if len(devs) > 1 else Context(prg, 11, 15, device=0, from_files=False)
            multi.set_opts(illumina=True, genome_size=60000)
            multi.set_threads(int(os.environ.get("STRESS_THREADS", "8")))
        else:
            multi.reset()
        multi.map_fastx(fq)
        got, got_prg = multi.coverage()
        if not (np.array_equal(got, want) and np.array_equal(got_prg, want_prg)):
            bad += 1
            print(f"devices {devs} round {r}: coverage differs: sum {int(got.sum())} vs {int(want.sum())} ({int(got.sum()) / int(want.sum()):.4f}), prg_reads {int(got_prg.sum())} vs {int(want_prg.sum())}, "
                  f"counters {multi.counters()} vs {single.counters()}", flush=True)
    print(f"devices {devs}: {rounds} rounds done, {bad} bad so far", flush=True)
shutil.rmtree(tmp)
sys.exit(1 if bad else 0)
