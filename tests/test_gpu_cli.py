"""Drop-in surface on the GPU: the two CLIs, the reference's four-symbol ABI (link.py) and
the looper.py contract (append, flush, stop on SIGTERM, per-process file index)."""
import ctypes
import json
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

from ataxxzero_amd import link, model, selfplay
from oracle import net_oracle
from oracle import oracle_lib as orc
from tests.helpers import replay_game_entry

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_abi_host_evaluated_selfplay(tmp_path):
    """The accelerated_generate_games.py host loop (:26-37, :54-83) against the GPU library:
    get_workload -> evaluate the (B,7,7,4) buffer on the host -> complete_workload."""
    B = 16
    conv, bn = model.random_init(1, 128, seed=6)
    out = str(tmp_path / "games.json")
    bufs = [np.zeros((B, 7, 7, 4), dtype=np.float32) for _ in (0, 1)]
    link.launch_threads(out.encode(), 10, ctypes.c_void_p(bufs[0].ctypes.data), ctypes.c_void_p(bufs[1].ctypes.data),
                        B, 2 * B)
    try:
        seen_rows = 0
        for _ in range(1600):
            w = link.get_workload()
            assert w in (0, 1)
            feats = bufs[w]
            assert set(np.unique(feats)) <= {0.0, 1.0}
            live = feats[:, :, :, 0].sum(axis=(1, 2)) == 49  # plane 0 is all ones on real rows
            seen_rows += int(live.sum())
            # rows carry the 4 blockers of the self-play start (cpp/self_play_client.cpp:23,198-200)
            assert (feats[live][:, :, :, 3].sum(axis=(1, 2)) == 4).all()
            p, v = net_oracle.forward(conv, bn, feats, dtype=np.float32)
            p = np.ascontiguousarray(p, dtype=np.float32)
            v = np.ascontiguousarray(v, dtype=np.float32)
            link.complete_workload(w, ctypes.c_void_p(p.ctypes.data), ctypes.c_void_p(v.ctypes.data))
        assert seen_rows > 1000
    finally:
        link.shutdown()
    lines = [l for l in open(out).read().split("\n") if l.strip()]
    assert len(lines) >= 3
    for line in lines:
        entry = json.loads(line)
        assert list(entry) == ["boards", "dists", "moves", "result"] and entry["result"] in (1, 2)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]
    # relaunch after shutdown works (shutdown clears state for another run, :740-749) and appends
    link.launch_threads(out.encode(), 4, ctypes.c_void_p(bufs[0].ctypes.data), ctypes.c_void_p(bufs[1].ctypes.data), B, 2 * B)
    link.shutdown()
    assert len([l for l in open(out).read().split("\n") if l.strip()]) == len(lines)


def test_accelerated_generate_games_cli_appends_and_stops_on_sigterm(tmp_path):
    conv, bn = model.random_init(1, 128, seed=3)
    net_path = str(tmp_path / "model-001.npy")
    model.save_model(net_path, conv, bn)
    games_path = str(tmp_path / "model-001-0.json")
    with open(games_path, "w") as f:
        f.write('{"pre-existing": true}\n')  # append mode must keep it (looper.py:24-30)
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", net_path,
                             "--output-games", games_path, "--visits", "8", "--buffer-size", "64"],
                            cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    deadline = time.time() + 240
    while time.time() < deadline:
        time.sleep(1.0)
        if sum(1 for l in open(games_path) if l.strip()) >= 30 or proc.poll() is not None:
            break
    assert proc.poll() is None, proc.stdout.read().decode()[-2000:]
    t0 = time.time()
    proc.send_signal(signal.SIGTERM)
    out, _ = proc.communicate(timeout=20)
    assert time.time() - t0 < 2.0, "looper.py waits 2 s before SIGKILL (looper.py:59-64)"
    text = out.decode()
    assert "Exiting cleanly" in text and proc.returncode == 0
    lines = [l for l in open(games_path) if l.strip()]
    assert lines[0].strip() == '{"pre-existing": true}' and len(lines) >= 31
    for line in lines[1:8]:
        entry = json.loads(line)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]
        # train.py:43-77 consumer assumptions: x moves on even plies, dists keyed by UAI strings
        assert all(isinstance(k, str) and len(k) in (2, 4) for d in entry["dists"] for k in d)


def test_generate_games_random_play_cli(tmp_path):
    path = str(tmp_path / "random.json")
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "generate_games.py"), "--random-play", "--output-games", path,
                          "--game-count", "2000"], cwd=ROOT, capture_output=True, timeout=600)   # BASELINE configs[0] verbatim
    wall = time.time() - t0
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert b"Doing random play" in res.stdout and b"Done generating games." in res.stdout
    lines = [l for l in open(path) if l.strip()]
    assert len(lines) == 2000
    rate = [l for l in res.stdout.decode().splitlines() if "Rate:" in l][-1]
    # beside BASELINE.md §2: the reference script, unmodified, one core of the build container: 74.5 s = 26.9 games/s
    note = ("config 1 (generate_games.py --random-play --game-count 2000): %s; whole process %.1f s = %.1f games/s; "
            "reference script 26.9 games/s" % (rate.split("] ", 1)[-1], wall, 2000 / wall))
    print(note)   # (the measurement proper is bench.py's config1_random_play leg; nothing is written outside tmp_path)
    plies = []
    for line in lines[:60]:
        assert '", "' not in line and "], [" in line  # json.dump default separators (generate_games.py:134)
        entry = json.loads(line)
        assert list(entry) == ["boards", "moves", "result"] and entry["result"] in (1, 2)
        # reference move values: ["c",[x,y]] / [[x0,y0],[x1,y1]] (ataxx_rules.py:108-127 via json)
        uai = []
        for m in entry["moves"]:
            sq = lambda xy: "abcdefg"[xy[0]] + str(7 - xy[1])
            uai.append(sq(m[1]) if m[0] == "c" else sq(m[0]) + sq(m[1]))
        e2 = {"boards": entry["boards"], "moves": uai}
        assert replay_game_entry(e2, orc.START_FEN_PLAIN) == entry["result"]
        plies.append(len(uai))
    assert 100 < np.mean(plies) < 260  # reference random play averages 182 plies
    # line for line the structure of what the reference's own generate_games.py writes (tests/golden/random_play_games.jsonl.gz)
    from tests.test_engine_fixtures_oracle import _reference_random_games, _shape
    ref_line = _reference_random_games()[0]
    ref = json.loads(ref_line)
    for line in lines[:20]:
        entry = json.loads(line)
        assert _shape(entry) == _shape(ref) and list(entry) == list(ref)
        assert json.dumps(entry) == line.rstrip("\n")                  # json.dump with its default separators, like ref_line
    assert json.dumps(ref) == ref_line


def test_two_half_batches_match_two_independent_engines():
    conv, bn = model.random_init(1, 128, seed=9)
    sp = selfplay.SelfPlay(conv, bn, games=64, visits=12, dtype="f32", seed=5, streams=2)
    sp.run(120)
    sp.sync()
    ref = []
    for i in range(2):
        e = link.Engine(selfplay.make_config(32, 12, seed=5 + 1000003 * i))
        e.run(sp.net, 120, link.DTYPE_F32)
        e.sync()
        ref.append(e)
    for a, b in zip(sp.engines, ref):
        for g in range(32):
            assert a.game_state(g).as_tuple() == b.game_state(g).as_tuple()
            for x, y in zip(a.tree(g), b.tree(g)):
                assert (x == y).all()
    sp.close()


def test_game_limit_is_split_over_uneven_half_batches():
    """SelfPlay.set_game_limit with several engines: batch i of K plays its own uids 0 .. ceil((N - i) / K) - 1, the limits
    add up to N, raising N later wakes idle slots, and batches may be of uneven size (65 games in 3: 22 + 22 + 21)."""
    conv, bn = model.random_init(1, 128, seed=9)
    sp = selfplay.SelfPlay(conv, bn, games=65, visits=6, dtype="f32", seed=5, streams=3, max_plies=60)
    assert [e.G for e in sp.engines] == [22, 22, 21]
    sp.set_emit_order(True)
    with pytest.raises(ValueError):
        sp.set_game_limit(2)
    lines = []
    for n_total in (100, 130):
        sp.set_game_limit(n_total)
        for _ in range(400):
            sp.run(50)
            lines += sp.drain()
            st = sp.stats()
            if st["games"] + st["dropped"] >= n_total:
                break
        assert st["games"] + st["dropped"] == n_total and len(lines) == st["games"]
        per_engine = [(n_total + 2 - i) // 3 for i in range(3)]
        assert sum(per_engine) == n_total
        for e, lim in zip(sp.engines, per_engine):
            assert all(e.game_state(g).phase == 3 and e.game_state(g).uid >= lim for g in range(e.G))
    sp.run(20)
    sp.sync()
    assert sp.stats() == st and sp.drain() == []
    sp.close()


def test_looper_iteration_generate_train_generate(tmp_path):
    """One iteration of looper.py's main loop (looper.py:117-153) with the two commands it issues:
    accelerated_generate_games.py on model-001, train.py -> model-002, generation again on model-002."""
    prefix = tmp_path / "run1"
    (prefix / "games").mkdir(parents=True)
    (prefix / "models").mkdir()
    conv, bn = model.random_init(1, 128, seed=11)
    m1, m2 = str(prefix / "models" / "model-001.npy"), str(prefix / "models" / "model-002.npy")
    model.save_model(m1, conv, bn)

    def generate(model_path, games_path, want):
        open(games_path, "a").close()  # looper.py:24-25 touches the file first
        proc = subprocess.Popen([sys.executable, "accelerated_generate_games.py", "--network", model_path,
                                 "--output-games", games_path, "--visits", "8", "--buffer-size", "128"],
                                cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        deadline = time.time() + 300
        while time.time() < deadline and proc.poll() is None:
            time.sleep(1.0)
            if sum(1 for l in open(games_path) if l.strip()) >= want:  # looper.py:5-12 count_games
                break
        proc.send_signal(signal.SIGTERM)
        out, _ = proc.communicate(timeout=30)
        assert proc.returncode == 0, out.decode()[-2000:]
        return sum(1 for l in open(games_path) if l.strip())

    g1 = str(prefix / "games" / "model-001-0.json")
    assert generate(m1, g1, 150) >= 150
    res = subprocess.run([sys.executable, "train.py", "--steps", "20", "--minibatch-size", "128", "--games", g1,
                          "--old-path", m1, "--new-path", m2], cwd=ROOT, capture_output=True, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    assert "Step:    0 -- loss:" in res.stdout.decode()
    conv2, bn2 = model.load_model(m2)
    assert any(not np.array_equal(a, b) for a, b in zip(conv, conv2))
    g2 = str(prefix / "games" / "model-002-0.json")
    assert generate(m2, g2, 60) >= 60
    entry = json.loads(open(g2).readline())
    assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]


def test_uai_interface_engine_plays_legal_moves(tmp_path):
    """The UAI protocol loop (uai_interface.py:41-88): handshake, per-ply `moves`, `go movetime`,
    `position fen`, showboard; the answers must be legal in the tracked position."""
    conv, bn = model.random_init(1, 128, seed=12)
    path = str(tmp_path / "model-001.npy")
    model.save_model(path, conv, bn)
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "uai_interface.py"), "--network-path", path, "--visits", "24"],
                            cwd=ROOT, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def send(s):
        proc.stdin.write(s + "\n")
        proc.stdin.flush()

    def read_until(prefix):
        while True:
            line = proc.stdout.readline()
            assert line, proc.stderr.read()[-2000:]
            if line.startswith(prefix):
                return line.strip()

    send("uai")
    assert read_until("uaiok") == "uaiok"
    send("isready")
    assert read_until("readyok") == "readyok"
    send("uainewgame")
    p = orc.pos_from_fen(orc.START_FEN_PLAIN)
    for ply in range(6):  # the ringmaster's flow: go -> bestmove -> `moves m` to every engine
        send("go movetime 100")
        best = read_until("bestmove ").split()[1]
        legal = [orc.move_string(m) for m in orc.movegen(p)]
        assert best in legal, (best, legal)
        send("moves " + best)
        c = orc.move_from_string(best)
        orc.lib().orc_makemove(p, c & 0xFF, c >> 8)
    send("showboard")
    read_until("boardok")
    send("position fen x5o/7/7/7/7/7/o5x o")
    send("go movetime 100")
    best = read_until("bestmove ").split()[1]
    q = orc.pos_from_fen("x5o/7/7/7/7/7/o5x o")
    assert best in [orc.move_string(m) for m in orc.movegen(q)]
    send("quit")
    proc.wait(timeout=30)
    # codecs keep the reference's names and conventions (uai_interface.py:34-39)
    sys.path.insert(0, ROOT)
    import uai_interface
    for s in ["f2", "c3d5"]:
        assert uai_interface.uai_encode_move(uai_interface.uai_decode_move(s)) == s
    for m in [("c", (4, 3)), ((4, 3), (2, 5))]:
        assert uai_interface.uai_decode_move(uai_interface.uai_encode_move(m)) == m


def test_generate_games_supervised_with_own_uai_engine_as_teacher(tmp_path):
    """generate_games.py --supervised CMD (generate_games.py:26-34,45-49): the teacher's move is the
    training move, the played move may be the opening randomisation's; entries have no dists."""
    conv, bn = model.random_init(1, 128, seed=13)
    net = str(tmp_path / "teacher.npy")
    model.save_model(net, conv, bn)
    out = str(tmp_path / "supervised.json")
    teacher = "%s %s --network-path %s --visits 6" % (sys.executable, os.path.join(ROOT, "uai_interface.py"), net)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "generate_games.py"), "--supervised", teacher,
                          "--supervised-ms", "50", "--output-games", out, "--game-count", "2"],
                         cwd=ROOT, capture_output=True, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in open(out) if l.strip()]
    assert len(lines) == 2
    for line in lines:
        entry = json.loads(line)
        assert list(entry) == ["boards", "moves", "result"] and entry["result"] in (1, 2)
        cells_of = lambda p: [int(v) for v in orc.board_cells(p)]
        p = orc.pos_from_fen(orc.START_FEN_PLAIN)
        for i, (b, m) in enumerate(zip(entry["boards"], entry["moves"])):
            assert cells_of(p) == b, i
            legal = [int(x) for x in orc.movegen(p)]
            sq = lambda xy: xy[0] + 7 * (6 - xy[1])
            mv = sq(m[1]) | (sq(m[1]) << 8) if m[0] == "c" else sq(m[0]) | (sq(m[1]) << 8)
            assert mv in legal, (i, m)
            if i + 1 < len(entry["boards"]):
                # the played move is the teacher's or a random legal one: the next board is a successor
                nxt = None
                for c in legal:
                    q = orc.Pos()
                    q.pieces[0], q.pieces[1], q.blockers, q.turn, q.ply = p.pieces[0], p.pieces[1], p.blockers, p.turn, p.ply
                    orc.lib().orc_makemove(q, c & 0xFF, c >> 8)
                    if cells_of(q) == entry["boards"][i + 1]:
                        nxt = q
                        break
                assert nxt is not None, i
                p = nxt
    # train.py consumes such entries through the one-hot branch (train.py:64-65)
    from ataxxzero_amd import training
    import random
    random.seed(1)
    entries = training.load_entries([out])
    f, pol, val = training.get_sample_from_entries(entries)
    assert abs(pol.sum() - 1.0) < 1e-6 and val[0] in (1, -1)


def test_accelerated_generate_games_extension_flags(tmp_path):
    """--one-random-move (the client's ONE_RANDOM_MOVE build as a switch), --select-budget, --streams 2, --eval-cache,
    --emit-order finish, --seed, --max-seconds."""
    conv, bn = model.random_init(1, 128, seed=5)
    net_path = str(tmp_path / "model-002.npy")
    model.save_model(net_path, conv, bn)
    games_path = str(tmp_path / "model-002-0.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", net_path,
                          "--output-games", games_path, "--visits", "6", "--buffer-size", "48", "--one-random-move",
                          "--select-budget", "3", "--streams", "2", "--eval-cache", "--emit-order", "finish", "--seed", "5",
                          "--max-seconds", "12"],
                         cwd=ROOT, capture_output=True, timeout=240)
    assert res.returncode == 0, res.stdout.decode()[-2000:] + res.stderr.decode()[-2000:]
    lines = [l for l in open(games_path) if l.strip()]
    assert len(lines) >= 10
    for line in lines[:12]:
        entry = json.loads(line)
        assert list(entry) == ["boards", "dists", "moves", "random_ply", "result"]
        assert 0 <= entry["random_ply"] < 120
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]


def test_generator_with_a_game_count_writes_exactly_those_games_and_exits(tmp_path):
    """--game-count N (extension; here through $AZH_GAME_COUNT, the way it reaches a generator that looper.py starts): N
    games in uid order, then a clean exit by itself."""
    conv, bn = model.random_init(1, 128, seed=5)
    net_path = str(tmp_path / "model-003.npy")
    model.save_model(net_path, conv, bn)
    games_path = str(tmp_path / "model-003-0.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", net_path,
                          "--output-games", games_path, "--visits", "6", "--buffer-size", "64",
                          "--seed", "9", "--max-seconds", "120"], cwd=ROOT, capture_output=True, timeout=400,
                         env=dict(os.environ, AZH_GAME_COUNT="70"))
    text = res.stdout.decode()
    assert res.returncode == 0, text[-2000:] + res.stderr.decode()[-2000:]
    totals = json.loads(next(l for l in text.splitlines() if l.startswith("Totals: "))[len("Totals: "):])
    lines = [l for l in open(games_path) if l.strip()]
    # 70 games WRITTEN: a game cut at 400 plies is made up for by one more (the limit is raised by what is missing)
    assert totals["games_in_flight"] == 70 and totals["written"] == len(lines) == totals["games"] == 70
    assert totals["seconds"] < 100
    for line in lines[:10]:
        entry = json.loads(line)
        assert replay_game_entry(entry, orc.START_FEN_SELFPLAY) == entry["result"]


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's keys, the roofline and the cpu_baseline objects."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--games", "96", "--visits", "8", "--blocks", "1",
                          "--steps", "30", "--warmup", "10", "--iters-per-step", "10", "--phase-fill", "30",
                          "--cpu-seconds", "1.5", "--no-target-leg"], cwd=ROOT, capture_output=True, timeout=400)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 10 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and "workload" in d["config"]
    assert abs(d["ms_per_step"] - 10 * d["ms_per_iteration"]) < 1e-6 * d["ms_per_step"]
    # games/s is the count of games handed to the host in the timed region over its wall time, nothing derived
    assert abs(d["games_per_s"] * d["ms_per_step"] * 30e-3 - d["games_finished_in_timed_region"]) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    halves = d["config"]["half_batches_in_flight"]
    assert halves == 2 and 0 < r["launches_timed"] <= r["launches_in_region"] == 300 * halves
    # two launches share the chip: the kernel's rate is taken over the region, one launch's own rate stands beside it
    assert abs(r["achieved"] - r["kernel_flops_in_region"] / r["region_s"] / 1e12) < 1e-6 * r["achieved"]
    assert r["per_launch"]["launches_sharing_the_chip"] == 2 and r["per_launch"]["avg_launch_ms"] == r["avg_launch_ms"]
    assert r["avg_launch_ms"] < d["ms_per_iteration"] and abs(r["per_launch_frac"] - r["per_launch"]["frac"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > 0   # null unless the committed PMC summary is of the current kernel source
    assert r["vendor_gemm_on_this_box"]["value"] > 100   # the library's bf16 GEMM on this box, measured beside the tower
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["null_evaluator_value"] > 0


def test_bench_line_carries_every_baseline_config_with_one_definition_of_the_tower_fraction():
    """BASELINE.json's configs[1], [3] and [4] are legs of the driver-run line (config2, config4, config5_arena; configs[0] =
    config1_random_play, configs[2] = the headline's shard), and in every leg `tower_frac_of_peak` is the chip-wide figure
    (all tower FLOPs of the leg's region / its wall time / peak) with `per_launch_frac` beside it.  Reduced game counts here;
    the legs' own sims/move, nets and dtypes."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--games", "96", "--visits", "8", "--blocks", "1",
                          "--steps", "4", "--warmup", "1", "--iters-per-step", "10", "--phase-fill", "20", "--no-cpu-baseline",
                          "--no-gemm-ceiling", "--legs", "one_batch,with_f16,config2,config4,config5_arena,config5_arena_balanced",
                          "--other-configs-games", "64", "--arena-ab"], cwd=ROOT, capture_output=True, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    d = json.loads([l for l in res.stdout.decode().splitlines() if l.strip()][-1])
    assert d["games_per_s_from_plies"] is None   # (only quoted for the 400-sim workload the snapshot was drawn from)
    from ataxxzero_amd import model
    for leg, visits, blocks, dtype in (("config2", 200, 12, "bf16"), ("config4", 800, 8, "f16")):
        c = d[leg]
        assert (c["visits"], c["net"], c["dtype"], c["games"], c["half_batches_in_flight"]) == (visits, "%dx128" % blocks, dtype, 64, 2)
        assert c["node_evals_per_s"] > 0 and c["nn_evals_per_s"] > 0
        tf = c["nn_evals_per_s"] * model.flops_per_eval(blocks, 128) / 1e12
        assert abs(c["tower_tflops"] - tf) < 1e-6 * tf and abs(c["tower_frac_of_peak"] - tf / 2500.0) < 1e-9
        assert c["per_launch_frac"] > 0 and abs(c["per_launch_frac"] - c["per_launch_tflops"] / 2500.0) < 1e-12
    for leg in ("one_batch", "with_f16"):
        assert "tower_frac_of_peak" in d[leg] and "per_launch_frac" in d[leg] and "tower_tflops_over_the_region" not in d[leg]
    a, b = d["config5_arena"], d["config5_arena_two_launches"]
    assert a["two_nets_in_one_launch"] is True and b["two_nets_in_one_launch"] is False
    for c in (a, b):
        assert c["games"] == 64 and c["wall_s"] > 0 and c["mcts_steps_per_s"] > 0 and c["tower_frac_of_peak"] > 0
        assert abs(c["games_per_s"] * c["wall_s"] - 64) < 1e-6
    assert a["score"] == b["score"] and a["mean_plies"] == b["mean_plies"]   # the same match, move for move
    # what the leg measures, said by the leg (round 6): when 99 % of the games were over, one launch's own rate, and — for a
    # match that starts with more than 512 games in flight — the rates while at least 512 were live; a second pairing of
    # two nets that are a match for each other
    bal = d["config5_arena_balanced"]
    assert bal["net_seeds"] == [27, 28] and a["net_seeds"] == [1, 2] and bal["games"] == 64
    for c in (a, b, bal):
        assert 0 < c["wall_s_until_99pct_games"] <= c["wall_s"] and 0 < c["iterations_until_99pct"] <= c["search_iterations"]
        assert c["per_launch_frac"] > 0 and abs(c["per_launch_frac"] - c["per_launch_tflops"] / 2500.0) < 1e-12
        assert c["tower_ms_per_launch"] < c["ms_per_iteration"] * 1.05
        assert "steps_per_s_while_ge512_live" not in c      # (64 games here: the phase does not exist)


def test_bench_py_two_ranks_on_one_gpu():
    """bench.py --gpus 2 started without a launcher: it spawns the two ranks itself (before any HIP call in the parent);
    both fold onto this box's one GPU (AZH_DEVICE_MOD=1) and meet over gloo (RCCL refuses two ranks on one device).
    Tiny workload: the launch path, the barriers, the per-rank seeds and the aggregation are what is tested."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(AZH_DEVICE_MOD="1", AZH_DIST_BACKEND="gloo")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--iters-per-step", "60", "--games", "256", "--visits", "16", "--blocks", "2",
                          "--phase-fill", "30", "--no-cpu-baseline", "--no-target-leg"], env=env, cwd=ROOT,
                         capture_output=True, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    out = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    # both ranks' work is in the aggregate: the two shards differ only in their seeds, so the job's total is about twice
    # what rank 0 counted itself
    total_steps = out["value"] * out["ms_per_step"] * 2e-3
    assert out["value"] > 0 and 1.6 < total_steps / out["counters"]["steps"] < 2.4
    assert total_steps < 1.05 * 2 * 256 * 120   # a game makes at most one step per iteration
    assert out["games_finished_in_timed_region"] == out["games_per_s"] * out["ms_per_step"] * 2e-3 or out["games_per_s"] > 0
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["launches_timed"] > 0


# A GPU box admits six processes on its card at once, this test process included: four ranks keep one slot in hand (a
# run that exceeds the limit is killed as a whole); five were run during development.  Four is the many-rank
# launch that can be rehearsed on the GPU (the eight-process launch itself: tests/test_distrib_gloo.py).
MANY_RANKS = 4


def test_bench_py_many_ranks_on_one_gpu():
    """The launch the round-end scaling run makes, with as many ranks as one GPU box admits: `bench.py --gpus 5` spawns
    its ranks, every rank builds its own engine and net on the (shared) GPU with its own Philox stream, they meet over
    gloo, rank 0 prints ONE line with the job's aggregate; a dying rank would fail the run (test_distrib_gloo.py)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(AZH_DEVICE_MOD="1", AZH_DIST_BACKEND="gloo")
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(MANY_RANKS), "--games", "512",
                          "--steps", "4", "--warmup", "1", "--iters-per-step", "60", "--visits", "32", "--blocks", "2",
                          "--phase-fill", "60", "--no-cpu-baseline", "--no-target-leg", "--no-gemm-ceiling"], env=env, cwd=ROOT,
                         capture_output=True, timeout=900)
    wall = time.time() - t0
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == MANY_RANKS and out["steps"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    total_steps = out["value"] * out["ms_per_step"] * 4e-3
    assert 0.8 * MANY_RANKS < total_steps / out["counters"]["steps"] < 1.2 * MANY_RANKS   # every rank's shard is in the sum
    assert total_steps <= MANY_RANKS * 512 * 240
    print("%d ranks on one GPU: %.1f s wall for a %.2f s timed region (import + rendezvous + engine set-up dominate)"
          % (MANY_RANKS, wall, out["ms_per_step"] * 4e-3))


def test_generator_processes_side_by_side_write_distinct_games(tmp_path):
    """looper.py --parallel-games-processes N (looper.py:33-41,70-74): N generator processes on model-%03i-{0..N-1}.json.
    The file index picks the GPU (folded onto this box's one) and the Philox stream; every file gets its own games."""
    import time
    conv, bn = model.random_init(2, 128, seed=6)
    net_path = str(tmp_path / "model-001.npy")
    model.save_model(net_path, conv, bn)
    paths = [str(tmp_path / ("model-001-%d.json" % i)) for i in range(MANY_RANKS)]
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "accelerated_generate_games.py"), "--network", net_path,
                               "--output-games", p, "--visits", "8", "--buffer-size", "64", "--seed", "100",
                               "--emit-order", "finish", "--max-seconds", "25"], cwd=ROOT, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for p in paths]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, text in zip(procs, outs):
        assert p.returncode == 0, text[-2000:]
        assert "Totals: " in text and "ring_overflow 0" in text
    games = [[l for l in open(p) if l.strip()] for p in paths]
    assert all(len(g) >= 20 for g in games), [len(g) for g in games]
    # distinct streams (seed + file index): two processes on one stream would write the same games.  Only games long
    # enough to be unlikely twice by chance count (at 8 visits a six-ply game does repeat): lines over 15 kB, 60+ plies
    long_games = [l for g in games for l in g if len(l) > 15000]
    assert len(long_games) > 100 and len(json.loads(long_games[0])["moves"]) >= 60
    assert len(set(long_games)) == len(long_games)
    seeds = [int(next(l for l in t.splitlines() if l.startswith("Philox seed:")).split(":")[1]) for t in outs]
    assert seeds == [100] * MANY_RANKS         # the same --seed everywhere: the process index is what separates them
    print("%d generator processes side by side: %.1f s wall, games per file %s" % (MANY_RANKS, time.time() - t0,
                                                                                     [len(g) for g in games]))


def test_bench_py_rank_over_rccl():
    """What every rank of an N > 1 run does, rehearsed with one rank (AZH_DIST_FORCE=1): torch's HIP runtime and the
    RCCL communicator come up first (init_process_group("nccl") + a device barrier), this package's library second,
    in the same process; the barriers and the reductions around the timed region go through RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "AZH_DIST_BACKEND")}
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env.update(AZH_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--iters-per-step", "60", "--games", "256", "--visits", "16", "--blocks", "2",
                          "--phase-fill", "30", "--no-cpu-baseline", "--no-target-leg"], env=env, cwd=ROOT,
                         capture_output=True, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["counters"]["steps"] > 0
    assert out["config"]["timing_barrier_backend"] == "nccl"
    assert 0 < out["roofline"]["frac"] < 1
