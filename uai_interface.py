#!/usr/bin/python
"""Drop-in for the reference's uai_interface.py: the UAI text-protocol engine (uai_interface.py:41-88)
over the GPU search, so the net can play any UAI master — including the reference's unmodified
uai_ringmaster.py — as `python uai_interface.py --network-path X.npy [--visits N]`.

Also exports the reference's move/square codecs (uai_interface.py:6-32) under the same names.
"""
import string
import sys


def uai_encode_square(xy):
    x, y = xy
    y = 6 - y
    return "%s%i" % (string.ascii_lowercase[x], y + 1)


def uai_encode_move(move):
    if move == "pass":
        return "0000"
    start, end = move
    if start == "c":
        return uai_encode_square(end)
    return "%s%s" % (uai_encode_square(start), uai_encode_square(end))


def uai_decode_square(s):
    x, y = string.ascii_lowercase.index(s[0].lower()), int(s[1]) - 1
    y = 6 - y
    return x, y


def uai_decode_move(s):
    if s in ("pass", "none", "0000"):
        return "pass"
    elif len(s) == 2:
        return "c", uai_decode_square(s)
    elif len(s) == 4:
        return uai_decode_square(s[:2]), uai_decode_square(s[2:])
    else:
        raise Exception("Bad UAI move string: %r" % s)


def main(args):
    from ataxxzero_amd import selfplay, uai
    selfplay.select_device(0)
    searcher = uai.Searcher(args.network_path, dtype=args.dtype, symmetry_average=args.symmetry_average)
    board = uai.Position.initial()
    while True:
        try:
            line = input()
        except EOFError:
            return
        if line == "quit":
            return
        elif line == "uai":
            print("id name AtaxxZero-MI355X")
            print("id author ataxxzero_amd")
            print("uaiok")
        elif line == "uainewgame":
            board = uai.Position.initial()
        elif line == "isready":
            print("readyok")
        elif line.startswith("moves "):
            for move in line[6:].split():
                board.move(uai.decode_move(move))
        elif line.startswith("position fen "):
            board = uai.Position.from_fen(line[13:])
            if args.show_game:
                print("===", file=sys.stderr)
                print(board, file=sys.stderr)
        elif line.startswith("go movetime "):
            ms = int(line[12:]) - args.safety_ms
            if args.visits is None:
                move = searcher.genmove(board, seconds=max(ms, 1) * 1e-3)
            else:
                move = searcher.genmove(board, visits=args.visits)
            print("info speed %f nps" % (searcher.last_steps / searcher.last_seconds,))
            print("bestmove %s" % (uai.encode_move(move),))
        elif line == "showboard":
            print(board)
            print("boardok")
        sys.stdout.flush()


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument("--network-path", metavar="NETWORK", type=str, help="Name of the model to load.")
    parser.add_argument("--visits", metavar="VISITS", default=None, type=int, help="Number of visits during MCTS.")
    parser.add_argument("--safety-ms", metavar="MS", default=0, type=int, help="Number of milliseconds to shave off of each movetime for safety.")
    parser.add_argument("--show-game", action="store_true", help="Show the game on stderr.")
    parser.add_argument("--symmetry-average", action="store_true", help="Evaluate every position as the mean over its 8 dihedral images (nn_evals.py:48-62; extension).")
    parser.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"], help="Tower arithmetic (extension).")
    args = parser.parse_args()
    print(args, file=sys.stderr)
    main(args)
