#!/usr/bin/python
"""Drop-in for the reference's uai_ringmaster.py on the path BASELINE.json scopes (config 5):
a head-to-head match between two AtaxxZero nets, played as ONE batched, device-resident match
on the GPU instead of UAI subprocess pairs.

    python uai_ringmaster.py --engine "python uai_interface.py --network-path A.npy --visits 400" \\
                             --engine "python uai_interface.py --network-path B.npy --visits 400" \\
                             [--pgn-out games.pgn] [--max-plies N] [--game-count 1000]

Same flags as the reference (uai_ringmaster.py:198-208); each --engine command is parsed for
`--network-path` and `--visits` (the two options uai_interface.py:97-100 takes).  Prints the
reference's `Wins: a - b (annulled: k)` line after every game and appends the reference's PGN
blocks.  Foreign UAI engines, --opening, --gauntlet with more than two engines and --show-games
need the subprocess ringmaster and are outside the GPU path.
"""
import argparse
import shlex
import sys

from ataxxzero_amd import arena, model, selfplay
from ataxxzero_amd.cli import flag, parse, switch


def parse_engine(cmd):
    p = argparse.ArgumentParser(add_help=False)
    p.add_argument("--network-path", type=str, required=True)
    p.add_argument("--visits", type=int, default=None)
    known, _ = p.parse_known_args(cmd)
    if known.visits is None:
        raise SystemExit("uai_ringmaster.py: the GPU arena needs --visits in every engine command "
                         "(time-controlled search depends on host speed)")
    return known.network_path, known.visits


if __name__ == "__main__":
    args = parse("Head-to-head match of two networks, both searched on one MI355X.", [
        flag("--engine", "engine command line (twice): python uai_interface.py --network-path X.npy --visits N",
             metavar="CMD", action="append"),
        switch("--show-games", "reference flag, not provided by the batched arena"),
        flag("--opening", "reference flag; only the empty opening is provided (random openings: --opening-depth)", metavar="MOVES"),
        flag("--opening-depth", "every pairing starts from its own opening of N uniformly random plies, played both ways — the "
                                "reference's get_opening with its module constant OPENING_DEPTH (uai_ringmaster.py:7,185-196) as a "
                                "flag.  Two near-deterministic engines otherwise replay the same few games; needs --game-count <= "
                                "the games in flight", type=int, default=0, metavar="N"),
        flag("--max-plies", "games longer than this are cut and annulled", type=int, metavar="N"),
        flag("--pgn-out", "PGN file the games are appended to", metavar="PGN"),
        switch("--gauntlet", "first engine against all others (with two engines: the same match)"),
        flag("--tc", "seconds per move, recorded in the PGN (the search is visit-limited)", type=float, default=1.0,
             metavar="SEC"),
        flag("--game-count", "size of the match (extension: the reference never stops): the games with uid below N are "
                             "played to completion and scored", type=int, default=1000, metavar="N"),
        flag("--concurrent", "games in flight on the GPU (extension).  Up to 512 the tower runs one board per workgroup from the "
                             "start, above that the 3-board tower until at most 512 games of the match are left: in 16 bits the two "
                             "kernels differ in the last bits, so the same --seed reproduces a match game for game only with the "
                             "same --concurrent and --game-count", type=int, metavar="N"),
        flag("--dtype", "tower arithmetic (extension).  Match play defaults to f16: with a trained net the f16 search picks the f32 "
                        "search's move in 100 %% of test positions, bf16 in 96 %% (DESIGN.md section 5); bf16 is 3-6 %% faster",
             default="f16", choices=["bf16", "f16", "f32"]),
        flag("--seed", "Philox seed (extension)", type=int, default=selfplay.DEFAULT_SEED),
    ])
    print("Options:", args)
    if not args.engine or len(args.engine) != 2:
        raise SystemExit("uai_ringmaster.py: the GPU arena plays exactly two --engine commands against each other")
    if args.show_games or (args.opening is not None and args.opening.strip()):
        raise SystemExit("uai_ringmaster.py: --show-games / non-empty --opening are not supported by the batched arena")
    engines = [tuple(shlex.split(e)) for e in args.engine]
    print("Engines:")
    for i, eng in enumerate(engines):
        print("%4i: %s" % (i + 1, eng))
    (path_a, visits_a), (path_b, visits_b) = parse_engine(engines[0]), parse_engine(engines[1])
    if visits_a != visits_b:
        raise SystemExit("uai_ringmaster.py: both engines must use the same --visits in the batched arena")
    selfplay.select_device(0)
    concurrent = args.concurrent or min(2048, args.game_count + args.game_count % 2)
    concurrent += concurrent % 2
    if args.opening_depth > 0 and args.game_count > concurrent:
        raise SystemExit("uai_ringmaster.py: --opening-depth needs the whole match in flight (--game-count %d > %d games; "
                         "raise --concurrent)" % (args.game_count, concurrent))
    match = arena.Match(model.load_model(path_a), model.load_model(path_b), visits_a, games=concurrent, dtype=args.dtype,
                        seed=args.seed,
                        # the reference stops a game once ply_number > --max-plies (uai_ringmaster.py:139-140): N + 1 moves
                        max_plies=args.max_plies + 1 if args.max_plies is not None else 400,
                        opening_depth=args.opening_depth)
    names = {"a": " ".join(engines[0]), "b": " ".join(engines[1])}
    wins = {"a": 0, "b": 0}
    annulled = 0
    written = 0
    # The match is a FIXED cohort: the games with uid < --game-count (slot g plays uids g, g + concurrent, ...; uid and
    # slot have the same parity, so the cohort holds every pairing both ways).  It runs until all of them have
    # finished.  Counting "the first N games to finish" instead would count the short games of a slot several times
    # while long games still in flight are thrown away — a net that wins quickly would look better than it is; the
    # reference ringmaster plays its games one after another to completion (uai_ringmaster.py:221-262).
    cohort = args.game_count
    match.set_game_limit(cohort)   # slots whose cohort games are over go idle: the batch thins out towards the end
    match.run(50)
    rounds = 0
    while written < cohort:
        match.fetch()              # the games finished so far ...
        # (every 16th round, between the fetch and the next run, where it waits for nothing: can the match still end?)
        ended = match.lost_games() if rounds % 16 == 15 else None
        rounds += 1
        match.run(50)              # ... are parsed, scored and written under the next iterations
        for game in sorted(match.drain(), key=lambda g: g["uid"]):
            if game["uid"] >= cohort:
                continue  # (cannot happen under the game limit; kept for callers that raise the limit)
            white = game["white"]
            black = "b" if white == "a" else "a"
            print('Game: "%s" vs "%s" with opening: [%s]' % (names[white], names[black], ", ".join(game["opening"])))
            if game["result"] in (1, 2):
                wins[white if game["result"] == 1 else black] += 1
            else:
                wins["a"] += 0.5
                wins["b"] += 0.5
                annulled += 1
            print("Wins: %s (annulled: %i)" % (" - ".join(str(wins[k]) for k in ("a", "b")), annulled))
            written += 1
            if args.pgn_out:
                arena.write_game_to_pgn(args.pgn_out, game, names[white], names[black], written, args.tc)
        sys.stdout.flush()
        if ended is not None and written < cohort:
            # every game of the cohort has ended on the device (or records were lost) and fewer were handed out: the
            # missing ones will never come — say so instead of searching an idle engine for ever
            match.close()
            print("uai_ringmaster.py: %s; %d of %d games were scored" % (ended, written, cohort), file=sys.stderr)
            sys.exit(3)   # (the generator CLI's exit code for lost records)
    match.close()
