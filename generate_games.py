#!/usr/bin/python
"""Drop-in for the reference's generate_games.py (generate_games.py:16-75, :127-140).  Same flags,
same entry format ({"boards", "moves", "result"}, json.dump default separators, one game per line).

* `--random-play`: bootstrap games, played by a HIP playout kernel instead of the pure-Python rules.
* `--supervised CMD [--supervised-ms N]`: teacher games — an external UAI engine picks every
  training move, with the reference's opening randomisation (ataxxzero_amd/supervised.py); rules on the GPU.

The reference's third mode, Python-MCTS self-play, is broken in the reference itself
(generate_games.py:43 calls a function that does not exist); use accelerated_generate_games.py
for network self-play.
"""
import json
import os
import time

from ataxxzero_amd import selfplay
from ataxxzero_amd.cli import flag, parse, switch

MAXIMUM_GAME_PLIES = 400

OPTIONS = [
    flag("--network", "unused by the two modes provided here (kept for the reference's command lines)", default="",
         metavar="NPY"),
    flag("--output-games", "file the games are written to (truncated first, as the reference does)", metavar="JSON"),
    flag("--group-index", "index of this process among parallel generators: picks the GPU and the RNG stream",
         type=int, default=0, metavar="N"),
    switch("--use-rpc", "reference flag, not provided (there is no RPC evaluator)"),
    switch("--random-play", "uniformly random games (bootstrap iteration)"),
    flag("--visit-count", "reference flag of its Python-MCTS mode; network self-play is accelerated_generate_games.py",
         type=int, default=200, metavar="N"),
    flag("--die-if-present", "stop as soon as this file exists", metavar="FILE"),
    switch("--show-game", "reference flag, not provided"),
    flag("--game-count", "stop after this many games", type=int, metavar="N"),
    switch("--no-write", "play but discard the games"),
    flag("--supervised", "command line of a UAI engine that picks the training moves", metavar="CMD"),
    flag("--supervised-ms", "movetime given to the teacher per move", type=int, default=100, metavar="MS"),
    flag("--seed", "Philox / host RNG seed (extension)", type=int, default=0),
]
args = parse("Bootstrap (random) and teacher (supervised) games in the .json format train.py reads.", OPTIONS)

if args.use_rpc or args.show_game or not (args.random_play or args.supervised):
    raise SystemExit("generate_games.py: --random-play and --supervised are provided by the MI355X build; "
                     "use accelerated_generate_games.py for network self-play.")
tag = "[%3i]" % args.group_index
if args.random_play:
    print("Doing random play! Loading no model, and not using RPC.")
selfplay.select_device(args.group_index)

if args.no_write:
    target = os.devnull
elif args.output_games:
    target = args.output_games
else:
    os.makedirs("games", exist_ok=True)
    target = os.path.join("games", "random-%s.json" % os.urandom(8).hex())
print(tag, "Writing to:", target)


class Sink:
    """The games file (truncated on open, generate_games.py:127), one json.dump per line, plus the stop rules."""

    def __init__(self, path):
        self.handle = open(path, "w")
        self.written = 0

    def accept(self, entry):
        """-> False once --game-count is reached."""
        if entry["result"] is None:
            print(tag, "Skipping game with null result.")
            return True
        self.handle.write(json.dumps(entry) + "\n")
        self.written += 1
        if args.game_count is not None and self.written >= args.game_count:
            print("Done generating games.")
            return False
        return True

    def should_stop(self):
        self.handle.flush()
        if args.die_if_present and os.path.exists(args.die_if_present):
            print("Exiting due to signal file!")
            return True
        return False


def teacher_games(sink):
    import random
    import shlex

    from ataxxzero_amd import supervised
    random.seed(args.seed * 1000003 + args.group_index)
    with supervised.Teacher(shlex.split(args.supervised)) as teacher:
        while True:
            entry = supervised.generate_game(teacher, args.supervised_ms)
            print(tag, "Generated a %i ply game with result %r." % (len(entry["boards"]), entry["result"]))
            if not sink.accept(entry) or sink.should_stop():
                return


def random_games(sink, batch=1024):
    for batch_index in range(1 << 30):
        left = None if args.game_count is None else max(args.game_count - sink.written, 1)
        entries = selfplay.random_play_entries(batch if left is None else min(batch, left),
                                               (args.seed << 20) + (args.group_index << 12) + batch_index,
                                               MAXIMUM_GAME_PLIES)
        more = all(sink.accept(entry) for entry in entries)   # all() stops at the first False: the count is exact
        mean_plies = sum(len(e["boards"]) for e in entries) / float(len(entries))
        print(tag, "Generated %i games (last batch mean %.1f plies)." % (sink.written, mean_plies))
        if not more or sink.should_stop():
            return


sink = Sink(target)
started = time.time()
try:
    (random_games if args.random_play else teacher_games)(sink)
finally:
    sink.handle.close()
    seconds = max(time.time() - started, 1e-9)
    # (the reference script prints no rate; BASELINE.md §2 measured it at 26.9 games/s for --random-play on one core)
    print(tag, "Rate: %.1f games/s (%i games in %.2f s)." % (sink.written / seconds, sink.written, seconds))
