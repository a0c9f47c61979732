#!/usr/bin/python
"""Drop-in for the reference's train.py (same flags; looper.py:140-148 calls it as
`python train.py --steps S --games G... --old-path A.npy --new-path B.npy`), on PyTorch-ROCm."""
import argparse

from ataxxzero_amd import training

if __name__ == "__main__":
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--games", metavar="PATH", required=True, nargs="+", help="Path to .json self-play games files.")
    parser.add_argument("--old-path", metavar="PATH", help="Path for input network.")
    parser.add_argument("--new-path", metavar="PATH", required=True, help="Path for output network.")
    parser.add_argument("--steps", metavar="COUNT", type=int, default=1000, help="Training steps.")
    parser.add_argument("--minibatch-size", metavar="COUNT", type=int, default=512, help="Minibatch size.")
    parser.add_argument("--learning-rate", metavar="LR", type=float, default=0.001, help="Learning rate.")
    parser.add_argument("--reference-bn-affine", action="store_true",
                        help="Train batch-norm gamma/beta like the reference and drop them on save (extension; see training.py).")
    args = parser.parse_args()
    print("Arguments:", args)
    training.train(args.games, args.old_path, args.new_path, steps=args.steps, minibatch_size=args.minibatch_size,
                   learning_rate=args.learning_rate, reference_bn_affine=args.reference_bn_affine)
