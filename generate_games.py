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
import argparse
import json
import os

from ataxxzero_amd import selfplay

MAXIMUM_GAME_PLIES = 400

parser = argparse.ArgumentParser(
    description="Tool for generating games in a .json format suitable for feeding into train.py.",
    formatter_class=argparse.ArgumentDefaultsHelpFormatter,
)
parser.add_argument("--network", metavar="PATH", default="", help="Path of the model to load.")
parser.add_argument("--output-games", metavar="PATH", type=str, default=None, help="Path to write .json games to.")
parser.add_argument("--group-index", metavar="N", default=0, type=int, help="Our index in the work group.")
parser.add_argument("--use-rpc", action="store_true", help="(reference flag; not supported here)")
parser.add_argument("--random-play", action="store_true", help="Generate games by totally random play.")
parser.add_argument("--visit-count", metavar="N", default=200, type=int, help="(reference flag; MCTS generation lives in accelerated_generate_games.py)")
parser.add_argument("--die-if-present", metavar="PATH", default=None, type=str, help="Die once a file is present at the target path.")
parser.add_argument("--show-game", action="store_true", help="(reference flag; not supported here)")
parser.add_argument("--game-count", metavar="N", default=None, type=int, help="Maximum number of games to generate.")
parser.add_argument("--no-write", action="store_true", help="Don't write out generated games at all.")
parser.add_argument("--supervised", metavar="CMD", default=None, type=str, help="Command for a UAI engine.")
parser.add_argument("--supervised-ms", metavar="N", default=100, type=int, help="Number of milliseconds per move for supervised generation.")
parser.add_argument("--seed", type=int, default=0, help="Philox seed (extension).")
args = parser.parse_args()

if args.use_rpc or args.show_game or not (args.random_play or args.supervised):
    raise SystemExit("generate_games.py: --random-play and --supervised are provided by the MI355X build; "
                     "use accelerated_generate_games.py for network self-play.")
if args.random_play:
    print("Doing random play! Loading no model, and not using RPC.")
selfplay.select_device(args.group_index)

output_path = "/dev/null" if args.no_write else args.output_games
if output_path is None:
    os.makedirs("games", exist_ok=True)
    output_path = os.path.join("games", "random-%s.json" % os.urandom(8).hex())
print("[%3i] Writing to: %s" % (args.group_index, output_path))

if not args.random_play:
    import random
    import shlex

    from ataxxzero_amd import supervised
    random.seed(args.seed * 1000003 + args.group_index)
    teacher = supervised.UAIPlayer(shlex.split(args.supervised))
    try:
        with open(output_path, "w") as f:
            games_generated = 0
            while True:
                entry = supervised.generate_game(teacher, args.supervised_ms)
                print("[%3i] Generated a %i ply game with result %r." % (
                    args.group_index, len(entry["boards"]), entry["result"]))
                if entry["result"] is None:
                    print("[%3i] Skipping game with null result." % (args.group_index,))
                    continue
                json.dump(entry, f)
                f.write("\n")
                f.flush()
                games_generated += 1
                if args.game_count is not None and games_generated >= args.game_count:
                    print("Done generating games.")
                    break
                if args.die_if_present and os.path.exists(args.die_if_present):
                    print("Exiting due to signal file!")
                    break
    finally:
        teacher.quit()
    raise SystemExit(0)

BATCH = 1024
with open(output_path, "w") as f:  # the reference truncates here (generate_games.py:127)
    games_generated = 0
    batch_index = 0
    done = False
    while not done:
        want = BATCH if args.game_count is None else min(BATCH, max(args.game_count - games_generated, 1))
        entries = selfplay.random_play_entries(want, (args.seed << 20) + (args.group_index << 12) + batch_index,
                                               MAXIMUM_GAME_PLIES)
        batch_index += 1
        for entry in entries:
            if entry["result"] is None:
                print("[%3i] Skipping game with null result." % (args.group_index,))
                continue
            json.dump(entry, f)
            f.write("\n")
            games_generated += 1
            if args.game_count is not None and games_generated >= args.game_count:
                print("Done generating games.")
                done = True
                break
        f.flush()
        print("[%3i] Generated %i games (last batch mean %.1f plies)." % (
            args.group_index, games_generated, sum(len(e["boards"]) for e in entries) / float(len(entries))))
        if args.die_if_present and os.path.exists(args.die_if_present):
            print("Exiting due to signal file!")
            break
