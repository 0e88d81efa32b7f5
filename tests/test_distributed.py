"""N>1 path on CPU: 2, 3 and 8 gloo ranks shard the reads, reduce the coverage vector, genotype on rank 0.

The device mapping itself cannot run here (no GPU, no CPU fallback), so each rank fills its shard's
coverage with the test oracle; what is under test is the product's sharding + reduce + set_coverage +
genotype plumbing (drprg_amd/distributed.py), i.e. that the N-rank result equals the 1-rank result."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from util import ROOT, cluster_fraction, map_params

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from util import Oracle, cluster_fraction, map_params
    from drprg_amd import Context, synth
    from drprg_amd.distributed import shard_batch, reduce_coverage

    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, k = 11, 15
    panel = synth.small_panel(seed=21)
    prg, genes = os.path.join(out, f"dr{{rank}}.prg"), os.path.join(out, f"genes{{rank}}.fa")
    panel.write(prg, genes)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=20000)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 5001, seed=9)       # odd count: uneven shards
    b, o = shard_batch(bases, offs, rank, world)
    md, er = map_params(k, True)
    orc = Oracle()
    covg, prg_reads, _ = orc.map_reads(b, o, orc.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    tc = torch.from_numpy(covg.view(np.int32).copy())
    tp = torch.from_numpy(prg_reads.view(np.int32).copy())
    try:  # (refused before any collective: every rank takes the same path)
        reduce_coverage(tc[::2], tp, 0)
        raise SystemExit("reduce_coverage accepted a non-contiguous tensor it cannot update in place")
    except ValueError:
        pass
    total = reduce_coverage(tc, tp, int(o[-1]))
    if rank == 0:
        ctx.set_coverage(tc.numpy().view(np.uint32), tp.numpy().view(np.uint32), total)
        ctx.genotype(genes, os.path.join(out, "multi.vcf"))
        np.save(os.path.join(out, "multi_covg.npy"), tc.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_shard_range_partitions():
    from drprg_amd.distributed import shard_range
    for n in (0, 1, 7, 100, 10_000_001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


import pytest


@pytest.mark.parametrize("world", [2, 3, 8])
def test_n_rank_reduce_equals_single_rank(tmp_path, oracle, world):
    """5001 reads over 2, 3 and 8 ranks (uneven shards: 8 ranks get 625 or 626 reads, 3 ranks 1667 each) -- BASELINE.json configs[3]'s
    layout, one process per rank over gloo: the reduced vector every rank-0 genotypes from is the single-process vector, and so is the
    VCF.  (Unmeasured on > 1 GPU: no multi-GPU box in the pool; this is the control flow.)"""
    from drprg_amd import Context, synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    import socket
    with socket.socket() as sock:  # a free port chosen at run time (concurrent jobs on one host must not collide)
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # single-process reference on the whole batch
    w, k = 11, 15
    panel = synth.small_panel(seed=21)
    prg, genes = str(tmp_path / "dr.prg"), str(tmp_path / "genes.fa")
    panel.write(prg, genes)
    ctx = Context(prg, w, k, device=-1, from_files=False)
    ctx.set_opts(illumina=True, genome_size=20000)
    gen = synth.HaplotypeGenomes(panel, genome_size=20000, n_hap=4, seed=3)
    bases, offs = synth.sample_short_reads(gen, 5001, seed=9)
    md, er = map_params(k, True)
    covg, prg_reads, _ = oracle.map_reads(bases, offs, oracle.build_index(panel.prgs, w, k), w, k, md, cluster_fraction(er, k), 10)
    assert np.array_equal(np.load(tmp_path / "multi_covg.npy"), covg)
    ctx.set_coverage(covg, prg_reads, int(offs[-1]))
    ctx.genotype(genes, str(tmp_path / "single.vcf"))
    strip = lambda p: [l for l in open(p) if not l.startswith("##fileDate")]
    assert strip(tmp_path / "multi.vcf") == strip(tmp_path / "single.vcf")


def test_reduce_coverage_refuses_tensors_it_cannot_update_in_place():
    """reduce_coverage writes the reduced values back into its arguments: a non-contiguous view or a tensor of another dtype would
    silently keep the unreduced values, so both are refused (single rank: nothing to reduce, nothing to check)."""
    import torch
    import torch.distributed as dist
    from drprg_amd.distributed import reduce_coverage
    a = torch.zeros(8, dtype=torch.int32)
    assert reduce_coverage(a, a[:2], 5) == 5  # no process group: a no-op
    if dist.is_available() and not dist.is_initialized():
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        try:
            assert reduce_coverage(a, a[:2], 7) == 7  # world size 1: still a no-op
        finally:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["torch", "native"])
def test_bench_ranks_whose_spin_ups_differ(comm):
    """bench.py --gpus 3 with every rank on the one GPU and gloo underneath (DRPRG_BENCH_BACKEND=gloo: control flow only).  The spin-up maps by the
    clock, so no two ranks run it equally often (forced here: rank r maps 120 ms x r longer), and collectives are matched by their order:
    a spin-up that reduced like the timed steps left the ranks waiting for each other until the transport's timeout (round 6, found by the
    profile run's eight ranks).  The line must come out, with every rank holding the sum of the ranks' vectors."""
    import json
    env = dict(os.environ, DRPRG_BENCH_BACKEND="gloo", DRPRG_BENCH_SPINUP_SKEW_MS="120", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--reads-per-gpu", "300000", "--steps", "3", "--warmup", "2", "--spinup-ms", "60",
           "--cpu-sample", "0", "--e2e", "0", "--comm", comm]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["config"]["all_ranks_hold_the_sum_of_the_ranks_vectors"] is True
    assert line["spinup"]["steps"] > 0 and line["value"] > 0
