#!/usr/bin/python
"""Trains a successor network from self-play game files on PyTorch-ROCm and writes it in the .npy layout.

Stands in for the reference's train.py: looper.py:140-148 runs
`python train.py --steps S --games G... --old-path A.npy --new-path B.npy`; the sample pipeline and optimiser
settings follow train.py:43-157 and live in ataxxzero_amd/training.py.
"""
from ataxxzero_amd import training
from ataxxzero_amd.cli import flag, parse, switch

OPTIONS = [
    flag("--games", "self-play .json files to sample positions from", metavar="JSON", nargs="+", required=True),
    flag("--old-path", "network to start from (.npy); random initialisation when absent", metavar="NPY"),
    flag("--new-path", "where the trained network is written (.npy)", metavar="NPY", required=True),
    flag("--steps", "optimiser steps", type=int, default=1000, metavar="N"),
    flag("--minibatch-size", "positions per step", type=int, default=512, metavar="N"),
    flag("--learning-rate", "momentum-SGD learning rate", type=float, default=0.001, metavar="LR"),
    switch("--reference-bn-affine", "train batch-norm gamma/beta as the reference does and drop them on save "
                                    "(extension; see training.py)"),
]

if __name__ == "__main__":
    args = parse(__doc__.splitlines()[0], OPTIONS)
    print("Arguments:", args)
    training.train(args.games, args.old_path, args.new_path, steps=args.steps, minibatch_size=args.minibatch_size,
                   learning_rate=args.learning_rate, reference_bn_affine=args.reference_bn_affine)
