#!/bin/bash
# train.py throughput on the GPU box: bootstrap games -> 200 training steps (run from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
T=$(mktemp -d)
python3 generate_games.py --random-play --output-games $T/games.json --game-count 400 > $T/gen.log 2>&1 || { tail -3 $T/gen.log; exit 1; }
t0=$(date +%s.%N)
python3 train.py --games $T/games.json --new-path $T/model-001.npy --steps ${STEPS:-200} > $T/train.log 2>&1 || { tail -5 $T/train.log; exit 2; }
t1=$(date +%s.%N)
tail -3 $T/train.log
python3 -c "print('train.py: %d steps of 512 in %.1f s = %.1f steps/s' % (${STEPS:-200}, $t1-$t0, ${STEPS:-200}/($t1-$t0)))"
rm -rf $T
