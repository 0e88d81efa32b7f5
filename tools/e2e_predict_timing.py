#!/usr/bin/env python3
"""Process start -> JSON timing of `drprg predict` (C++ front end) on the SURVEY 8d index with synthetic 150 bp reads, plain
FASTQ in /dev/shm, on the GPU box: a wild-type sample and a sample with one off-panel SNP (second mapping pass).
Usage: python tools/e2e_predict_timing.py [n_reads] [threads]"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from drprg_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
DS = os.path.join(ROOT, "tests", "golden", "downstream")
tmp = tempfile.mkdtemp(prefix="drprg_predict_", dir="/dev/shm")
idx = os.path.join(tmp, "idx")
os.makedirs(os.path.join(idx, "msas"))
for f in (".config.toml", "genes.fa", "genes.fa.fai", "panel.bcf", "panel.bcf.csi", "rules.csv"):
    shutil.copy(os.path.join(DS, f), os.path.join(idx, f))
panel = synth.mtb_8d_panel(DS)
panel.write(os.path.join(idx, "dr.prg"))
exe = os.path.join(ROOT, "drprg_amd", "bin")
subprocess.run([os.path.join(exe, "pandora"), "index", "-t", "8", "-w", "11", "-k", "15", os.path.join(idx, "dr.prg")], check=True, stdout=subprocess.DEVNULL)
g = panel.names.index("katG")
for label in ("wild_type", "off_panel_snp"):
    trees = list(panel.trees)
    refs = list(panel.refs)
    if label == "off_panel_snp":
        p = 100 + 3 * 300 + 1
        refs[g] = refs[g][:p] + ("A" if refs[g][p] != "A" else "C") + refs[g][p + 1:]
    # background genome with the first-allele loci implanted (the mutated katG for the second sample)
    rng = np.random.default_rng(4411532)
    total = sum(len(r) for r in refs)
    spacer = (synth.MTB_GENOME_SIZE - total) // (len(refs) + 1)
    genome = "".join(synth.random_seq(rng, spacer) + r for r in refs) + synth.random_seq(rng, spacer)
    class G:  # the two attributes sample_short_reads needs
        haps = [np.frombuffer(genome.encode(), np.uint8)]
        lens = np.array([len(genome)], dtype=np.int64)
    bases, offs = synth.sample_short_reads(G, n, seed=2)
    fq = os.path.join(tmp, label + ".fq")
    synth.write_fastq_fixed(fq, bases, 150)
    out = os.path.join(tmp, "out_" + label)
    # E2E_REPS=N: N runs per variant (min / median / max); E2E_VARIANTS="name:ENV=v,ENV=v;name2:..." runs each set of environment
    # variables on the same file (an A/B on one box: boxes differ by more than most changes do)
    reps = int(os.environ.get("E2E_REPS", "1"))
    variants = [("default", {})]
    for spec in filter(None, os.environ.get("E2E_VARIANTS", "").split(";")):
        name, _, kv = spec.partition(":")
        variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    import re
    for vname, venv in variants:
        totals, exits = [], []
        for rep in range(reps):
            shutil.rmtree(out, ignore_errors=True)
            t = time.time()
            r = subprocess.run([os.path.join(exe, "drprg"), "predict", "-x", idx, "-i", fq, "-o", out, "-s", label, "-I", "-t", threads, "-v"], capture_output=True,
                               text=True, env=dict(os.environ, DRPRG_HIP_T0=repr(t), **venv))
            dt = time.time() - t
            assert r.returncode == 0, r.stderr
            wrote = [float(m.group(1)) for m in re.finditer(r"\+([0-9.]+)s\] wrote", r.stderr)]
            totals.append(dt)
            exits.append(dt - wrote[-1] if wrote else float("nan"))
        res = json.load(open(os.path.join(out, label + ".drprg.json")))
        calls = {d: v["predict"] for d, v in res["susceptibility"].items() if v["predict"] != "S"}
        stat = lambda v: f"{min(v):.3f} / {sorted(v)[len(v) // 2]:.3f} / {max(v):.3f}"
        print(f"{label} [{vname}]: {n} reads ({os.path.getsize(fq) / 1e9:.2f} GB FASTQ), process start -> exit, {reps} run(s), min / median / max: {stat(totals)} s "
              f"= {n / sorted(totals)[len(totals) // 2] / 1e6:.1f} M reads/s; of that after the JSON was on disk: {stat(exits)} s; non-S: {calls}", flush=True)
    dt = totals[-1]
    print("   " + " | ".join(l for l in r.stderr.splitlines() if "discover" in l or "novel" in l), flush=True)
    if os.environ.get("E2E_STDERR"):
        print(r.stderr, flush=True)
    if os.environ.get("E2E_ROCPROF") and label == "off_panel_snp":
        # the same command once more under rocprofv3 (kernel trace): where the device time of the whole prediction goes
        import csv, glob
        pdir = os.path.join(tmp, "prof")
        shutil.rmtree(out, ignore_errors=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", pdir, "-o", "cli", "--", os.path.join(exe, "drprg"), "predict", "-x", idx,
                        "-i", fq, "-o", out, "-s", label, "-I", "-t", threads], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", DRPRG_HIP_SLOW_EXIT="1"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in glob.glob(os.path.join(pdir, "**", "*kernel_stats.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
            print("   kernels of the whole command (rocprofv3 --kernel-trace --stats): name, calls, total ms, average us")
            for r in rows[:14]:
                print(f"   {r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['AverageNs']) / 1e3:9.1f}", flush=True)
    if label == "wild_type" and os.environ.get("E2E_GZ", "1") != "0":
        # the same sample as one plain gzip stream (what `gzip` writes; inflated by all -t threads, csrc/pgunzip.cpp)
        gz = fq + ".gz"
        with open(gz, "wb") as fh:
            subprocess.run(["gzip", "-1", "-c", fq], stdout=fh, check=True)
        out = os.path.join(tmp, "out_gz")
        for th in (threads, "32"):
            shutil.rmtree(out, ignore_errors=True)
            t = time.time()
            r = subprocess.run([os.path.join(exe, "drprg"), "predict", "-x", idx, "-i", gz, "-o", out, "-s", "gz", "-I", "-t", th], capture_output=True, text=True)
            dt = time.time() - t
            assert r.returncode == 0, r.stderr
            same = json.load(open(os.path.join(out, "gz.drprg.json")))["susceptibility"] == res["susceptibility"]
            print(f"wild_type as .fq.gz ({os.path.getsize(gz) / 1e9:.2f} GB), -t {th}: process start -> JSON {dt:.2f}s = {n / dt / 1e6:.1f} M reads/s; "
                  f"same calls as from plain text: {same}", flush=True)
        os.remove(gz)
    os.remove(fq)
shutil.rmtree(tmp)
