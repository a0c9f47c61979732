#!/usr/bin/python
"""UAI engine over the GPU search: `python uai_interface.py --network-path X.npy [--visits N]`.

Stands in for the reference's script of the same name (command set uai_interface.py:41-88, options :97-100), so
any UAI master — the reference's unmodified uai_ringmaster.py included — can drive the net.  The dialogue itself
lives in ataxxzero_amd/uai.py (Session); the four codec helpers other reference modules import from here
(train.py:64-69, uai_ringmaster.py:42,59) are re-exported under their reference names.
"""
import argparse
import sys

from ataxxzero_amd.uai import square_to_xy as uai_decode_square
from ataxxzero_amd.uai import text_to_xy_move as uai_decode_move
from ataxxzero_amd.uai import xy_move_to_text as uai_encode_move
from ataxxzero_amd.uai import xy_to_square as uai_encode_square

__all__ = ["uai_encode_square", "uai_encode_move", "uai_decode_square", "uai_decode_move", "main"]


def main(options):
    from ataxxzero_amd import selfplay, uai
    selfplay.select_device(0)
    searcher = uai.Searcher(options.network_path, dtype=options.dtype, symmetry_average=options.symmetry_average)
    session = uai.Session(searcher, visits=options.visits, safety_ms=options.safety_ms, show_game=options.show_game,
                          log=sys.stderr)
    session.serve(sys.stdin, sys.stdout)


if __name__ == "__main__":
    cli = argparse.ArgumentParser(description="UAI engine: MCTS and the policy/value net on one MI355X.")
    cli.add_argument("--network-path", required=True, metavar="NPY", help=".npy weight file in the model.py layout")
    cli.add_argument("--visits", type=int, default=None, metavar="N", help="search exactly N steps per move instead of using the clock")
    cli.add_argument("--safety-ms", type=int, default=0, metavar="MS", help="margin subtracted from every movetime")
    cli.add_argument("--show-game", action="store_true", help="echo positions set by `position fen` to stderr")
    cli.add_argument("--symmetry-average", action="store_true", help="evaluate every position as the mean over its 8 dihedral images (nn_evals.py:48-62; extension)")
    cli.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32"],
                     help="tower arithmetic (extension).  Match play defaults to f16: with a trained net the f16 search picks "
                          "the f32 search's move in 100 %% of test positions, bf16 in 96 %% (DESIGN.md section 5); bf16 is 3-6 %% faster")
    options = cli.parse_args()
    print(options, file=sys.stderr)
    main(options)
