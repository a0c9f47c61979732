"""N > 1 path on CPU: two ranks over gloo exercise the sharding / timing plumbing bench.py
uses (no data-path collective exists: games are independent shards)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    from ataxxzero_amd import distrib
    g = distrib.Group(backend="gloo")
    assert g.world == 2 and g.rank in (0, 1)
    g.barrier()
    seed = distrib.shard_seed(20260101, g.rank)
    units, secs = (1000.0, 2.0) if g.rank == 0 else (3000.0, 4.0)
    total, t, rate = distrib.aggregate(g, units, secs)
    g.barrier()
    print(json.dumps({"rank": g.rank, "seed": seed, "total": total, "t": t, "rate": rate}))
    g.close()
""") % ROOT


def test_two_ranks_aggregate_sum_over_max_time(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err.decode()[-2000:]
        outs.append(json.loads(out.decode().strip().splitlines()[-1]))
    seeds = sorted(o["seed"] for o in outs)
    assert seeds == [20260101, 20260102]  # distinct Philox stream per rank
    for o in outs:
        assert o["total"] == 4000.0 and o["t"] == 4.0 and o["rate"] == 1000.0  # units summed / max time


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from ataxxzero_amd import distrib
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        g = distrib.Group()
        assert g.world == 1 and g.dist is None
        g.barrier()
        assert distrib.aggregate(g, 10.0, 2.0) == (10.0, 2.0, 5.0)
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v


def test_bench_py_starts_its_own_ranks_and_aggregates():
    """`python bench.py --gpus 2` with no launcher: bench.py itself starts one process per rank (before any HIP call),
    the ranks meet over gloo, rank 0 prints the one JSON line.  --plumbing-selftest replaces the GPU work by fixed
    units (rank r: 1000 (r + 1) units in 1 + r seconds), so the launch / barrier / sum-over-max-time path runs on a
    CPU-only box; the GPU-side twin of this test is tests/test_gpu_cli.py::test_bench_py_two_ranks_on_one_gpu."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest"],
                         env=env, capture_output=True, timeout=240)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    out = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["value"] is None
    assert out["units_total"] == 3000.0 and out["t_max"] == 2.0 and out["rate"] == 1500.0
    assert out["seeds"] == [20260101, 20260102]
    # each rank's own figures beside the aggregate: a straggler shows (rank 1 took twice as long), and so would two ranks
    # on one card
    pr = out["per_rank"]
    assert pr["value"] == [1000.0, 1000.0] and pr["ms_per_step"] == [1000.0, 2000.0]
    assert pr["ms_per_step_min_max"] == [1000.0, 2000.0] and pr["value_min_max"] == [1000.0, 1000.0]
    assert pr["device"] == [0, 1] and pr["pci_bus_id"] == ["0000:05:00.0", "0000:06:00.0"]
    assert pr["ranks_sharing_a_device"] == 0
    # the ranks' host threads sit on disjoint core shares (when this box has at least two cores to share out)
    shares = [tuple(int(x) for x in h.split(" ")[0].split("-")) for h in pr["host_cores"]]
    if len(os.sched_getaffinity(0)) >= 2:
        assert shares[0][1] < shares[1][0]

    env["AZH_SELFTEST_SAME_CARD"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest"],
                         env=env, capture_output=True, timeout=240)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert json.loads(res.stdout.decode().strip().splitlines()[-1])["per_rank"]["ranks_sharing_a_device"] == 1


def test_core_shares_are_disjoint_and_cover_every_rank():
    sys.path.insert(0, ROOT)
    from ataxxzero_amd import distrib
    cores = list(range(3, 3 + 64))
    shares = [distrib.core_share(r, 8, cores) for r in range(8)]
    assert all(len(s) == 8 for s in shares) and sorted(sum(shares, [])) == cores
    assert distrib.core_share(0, 1, cores) == cores                         # one rank keeps everything (the CPU baseline needs it)
    assert distrib.core_share(5, 8, [0, 1, 2]) == [0, 1, 2]                 # fewer cores than ranks: nobody is starved
    assert distrib.numbers_to_pci(distrib.pci_to_numbers("0000:f5:00.0")) == "0000:f5:00.0"
    assert distrib.numbers_to_pci(distrib.pci_to_numbers("garbage")) is None


def test_bench_py_eight_ranks_as_the_scaling_run_starts_them():
    """The rank count the round-end scaling run ends with: `bench.py --gpus 8` starts eight processes, eight distinct Philox
    seeds, one aggregate line with n_gpus 8 (units summed, max-over-ranks time).  The GPU work is replaced by fixed
    units: a GPU box admits six processes on its card, so the eight-process launch itself is rehearsed here, and the
    GPU side of a many-rank launch with four ranks in tests/test_gpu_cli.py."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-selftest"],
                         env=env, capture_output=True, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["units_total"] == 1000.0 * 36 and out["t_max"] == 8.0
    assert out["seeds"] == [20260101 + r for r in range(8)]
    assert out["per_rank"]["device"] == list(range(8)) and len(set(out["per_rank"]["pci_bus_id"])) == 8
    assert out["per_rank"]["ms_per_step_min_max"] == [1000.0, 8000.0]
    print("8-rank launch, rendezvous and teardown: %.1f s wall" % (time.time() - t0))


def test_bench_py_under_the_launcher_the_driver_uses():
    """The round-end scaling run starts bench.py as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: the ranks read RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from the launcher and must not start ranks of their own; rank 0 alone prints the line."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest"],
                         env=env, capture_output=True, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["units_total"] == 3000.0 and out["t_max"] == 2.0
    assert out["seeds"] == [20260101, 20260102]


def test_bench_py_fails_fast_when_a_rank_dies():
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["AZH_SELFTEST_FAIL_RANK"] = "1"
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest"],
                         env=env, capture_output=True, timeout=120)
    assert res.returncode != 0 and b"rank 1 -> exit 3" in res.stderr
    assert time.time() - t0 < 60   # rank 0 was not left waiting at the barrier


def test_bench_py_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest"],
                         env=env, capture_output=True, timeout=120)
    assert res.returncode != 0 and b"--gpus 2 but WORLD_SIZE=1" in res.stderr


def test_cpu_baseline_threads_play_with_a_null_evaluator():
    # bench.py's cpu_baseline target (the reference's thread-per-game architecture): launch, batch hand-off, shutdown
    sys.path.insert(0, ROOT)
    from ataxxzero_amd import build
    from tools.cpu_baseline import driver
    build.build_cpu_baseline()
    r = driver.run(None, visits=50, buffer_entries=8, seconds=1.0, warmup_seconds=0.3)
    assert r["threads"] == 16 and r["steps_per_s"] > 1000 and r["plies_per_s"] > 5
    assert 0.9 <= r["steps_per_s"] / r["evals_per_s"] <= 1.2   # ~one evaluation per step, plus terminal re-hits
