"""The host loop of accelerated_generate_games.py on the CPU: the script itself, run in a child process, with the engine behind
`selfplay.SelfPlay` replaced by a recording stand-in (no device, no library).  What is checked is the order of the calls —
since round 3 the next run is enqueued BEFORE the finished games of the last one are formatted and written — and that no game
is written twice or lost on the ways out of the loop (SIGTERM as looper.py sends it, --max-seconds, --game-count).
"""
import json
import os
import signal
import subprocess
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, os, runpy, sys, time
sys.path.insert(0, %(root)r)
from ataxxzero_amd import model, selfplay

calls = []
LOG = %(log)r


class Net:
    blocks, filters = 12, 128


class FakeSelfPlay:
    """Every run "finishes" three games; they become visible to the host at the next fetch.  fetch / drain mirror the product:
    `drain` is link.Engine.drain_json (it calls the library's drain until a call hands out no game) and `_drain_call` is
    azh_engine_drain_json's state machine (csrc/engine.hip): a call with nothing pending and nothing staged fetches by
    itself when work has been enqueued since the last fetch — and so waits for whatever run is enqueued — unless an
    explicit fetch covers the drain sequence."""

    def __init__(self, conv, bn, games, visits, **kw):
        self.net = Net()
        self.games = games
        self.in_flight = []      # games of the runs enqueued and not yet fetched
        self.staged = []         # fetched, not yet formatted
        self.covers = False      # an explicit fetch covers the drain sequence that follows it
        self.unfetched_work = True   # a run has been enqueued since the last fetch
        self.old_semantics = bool(os.environ.get("FAKE_ROUND3_DRAIN"))   # round 3's library: every empty call fetched
        self.next_game = 0
        self.limit = None
        self.counters = dict.fromkeys(["steps", "nn_evals", "levels", "children", "new_moves", "plies", "games", "dropped",
                                       "edge_overflow", "reroot_nodes", "reroot_edges", "ring_overflow", "cache_hits",
                                       "parked", "reroot_spills"], 0)
        calls.append(["create", games, visits])
        calls.append(["seed", kw.get("seed")])

    def set_emit_order(self, by_uid):
        calls.append(["emit_order", bool(by_uid)])

    def set_game_limit(self, n):
        self.limit = n
        calls.append(["limit", n])

    def set_thin_batches(self, mode):
        calls.append(["thin", mode])

    def run(self, iterations):
        calls.append(["run", iterations])
        self.unfetched_work = True
        for _ in range(3):
            if self.limit is None or self.next_game < self.limit:
                self.in_flight.append(self.next_game)
                self.next_game += 1
        self.counters["nn_evals"] += 1000 * iterations
        self.counters["steps"] += 1000 * iterations

    def _fetch_records(self):
        time.sleep(0.01)         # the wait for the GPU: everything enqueued so far has to end
        self.unfetched_work = False
        self.staged += self.in_flight
        self.counters["games"] += len(self.in_flight)
        self.in_flight = []

    def fetch(self):
        calls.append(["fetch"])
        self._fetch_records()
        self.covers = not self.old_semantics

    def _drain_call(self):
        if not self.staged and not self.covers and (self.unfetched_work or self.old_semantics):
            calls.append(["fetch_in_drain"])
            self._fetch_records()
        out, self.staged = self.staged, []
        if not out:
            self.covers = False
        return out

    def drain(self):
        lines = []
        while True:
            got = self._drain_call()
            if not got:
                break
            lines += got
        calls.append(["drain", len(lines)])
        return [json.dumps({"game": g}).encode() for g in lines]

    def stats(self):
        calls.append(["stats"])
        return dict(self.counters)

    def close(self):
        calls.append(["close"])
        with open(LOG, "w") as f:
            json.dump(calls, f)


selfplay.SelfPlay = FakeSelfPlay
selfplay.select_device = lambda index: calls.append(["device_for_index", index]) or index
model.load_model = lambda path: ([], [])
sys.argv = ["accelerated_generate_games.py"] + %(argv)r
runpy.run_path(os.path.join(%(root)r, "accelerated_generate_games.py"), run_name="__main__")
'''


def launch(tmp_path, argv, env=None, name="games-0.json"):
    out = str(tmp_path / name)
    log = str(tmp_path / "calls.json")
    code = STUB % {"root": ROOT, "log": log, "argv": ["--network", "none.npy", "--output-games", out] + argv}
    e = dict(os.environ)
    e.pop("AZH_GAME_COUNT", None)
    e.pop("AZH_SEQUENTIAL_DRAIN", None)
    e.update(env or {})
    proc = subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return proc, out, log


def finish(proc, out, log):
    text = proc.communicate(timeout=60)[0].decode()
    games = [json.loads(l)["game"] for l in open(out)] if os.path.exists(out) else []
    calls = json.load(open(log)) if os.path.exists(log) else None
    return text, games, calls


def loop_calls(calls):
    return [c[0] for c in calls if c[0] in ("run", "fetch", "fetch_in_drain", "drain")]


def test_next_run_is_enqueued_before_the_games_are_formatted(tmp_path):
    proc, out, log = launch(tmp_path, ["--max-seconds", "0.3"])
    text, games, calls = finish(proc, out, log)
    assert proc.returncode == 0, text
    seq = loop_calls(calls)
    assert seq[:4] == ["run", "fetch", "run", "drain"]
    # steady state: fetch, run, drain — and NO call of the drain sequence waits for the device (a fetch inside the drain
    # would wait for the run that was just enqueued: the order would be sequential again without saying so)
    assert "fetch_in_drain" not in seq
    body = seq[1:]
    rounds = len(body) // 3
    assert rounds >= 3
    assert body[:3 * (rounds - 1)] == ["fetch", "run", "drain"] * (rounds - 1)
    # the way out: the run still in flight is fetched explicitly, so every finished game is written exactly once
    assert seq[-2:] == ["fetch", "drain"]
    assert games == list(range(len(games))) and len(games) == 3 * seq.count("run")
    assert "Totals: " in text and "all game slots shutdown." in text


def test_the_stand_in_would_have_caught_round_3s_implicit_fetch(tmp_path):
    """With round 3's library semantics (a drain call that finds nothing staged always fetches) the loop's last drain call of
    every round waits for the run that was just enqueued: the stand-in shows it, so the assertion above is not vacuous."""
    proc, out, log = launch(tmp_path, ["--max-seconds", "0.2"], env={"FAKE_ROUND3_DRAIN": "1"})
    text, games, calls = finish(proc, out, log)
    assert proc.returncode == 0, text
    seq = loop_calls(calls)
    assert seq[:4] == ["run", "fetch", "run", "fetch_in_drain"]
    assert games == list(range(len(games)))         # (correct games all the same: that is why nobody noticed)


def test_sequential_order_on_request_and_with_a_game_target(tmp_path):
    proc, out, log = launch(tmp_path, ["--max-seconds", "0.2"], env={"AZH_SEQUENTIAL_DRAIN": "1"})
    text, games, calls = finish(proc, out, log)
    assert proc.returncode == 0, text
    seq = loop_calls(calls)
    assert "fetch" not in seq and seq[:3] == ["run", "fetch_in_drain", "drain"]   # sequential: the drain itself waits
    assert games == list(range(3 * seq.count("run")))

    sub = tmp_path / "target"
    sub.mkdir()
    proc, out, log = launch(sub, ["--game-count", "10"])
    text, games, calls = finish(proc, out, log)
    assert proc.returncode == 0, text
    seq = loop_calls(calls)
    assert "fetch" not in seq                       # with a target the rounds stay sequential (counters read every round)
    assert ["limit", 10] in calls and ["create", 10, 100] in calls
    assert games == list(range(10))                 # exactly the games below the limit, then the script ends by itself


def test_sigterm_as_looper_sends_it_ends_the_run_within_its_two_seconds(tmp_path):
    proc, out, log = launch(tmp_path, [])
    deadline = time.time() + 20
    while time.time() < deadline and not (os.path.exists(out) and os.path.getsize(out) > 200):
        time.sleep(0.05)
    t0 = time.time()
    proc.send_signal(signal.SIGTERM)                # looper.py:57-64: SIGTERM, then kill after 2 s
    text, games, calls = finish(proc, out, log)
    assert time.time() - t0 < 2.0
    assert proc.returncode == 0, text
    seq = loop_calls(calls)
    assert seq[-1] == "drain" and calls[-1] == ["close"]
    assert games == list(range(len(games))) and len(games) == 3 * seq.count("run")


def test_looper_style_process_index_picks_device_and_seed(tmp_path):
    """looper.py:70-74 starts N generators on games/model-%03i-%i.json: the trailing index is the GPU the process takes and
    the offset of its Philox seed (one process per GPU, distinct streams, no collective)"""
    seen = []
    for index in (0, 3, 7):
        sub = tmp_path / str(index)
        sub.mkdir()
        proc, out, log = launch(sub, ["--max-seconds", "0.05", "--seed", "100"], name="model-004-%d.json" % index)
        text, games, calls = finish(proc, out, log)
        assert proc.returncode == 0, text
        assert ["device_for_index", index] in calls and ["seed", 100 + index] in calls
        seen.append(index)
    assert seen == [0, 3, 7]
