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
