#!/bin/bash
# The generator under a game target, with and without thin batches for its last games (one board per workgroup once at most
# 512 games are left): wall time of `accelerated_generate_games.py --game-count N`, same seed, one box, one call.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
N=${1:-1000}
python3 -c "
from ataxxzero_amd import model
conv, bn = model.random_init(12, 128, seed=1)
model.save_model('/tmp/thin-ab.npy', conv, bn)"
for mode in ${MODES:-thin three_boards thin three_boards}; do
  rm -f /tmp/thin-ab-$mode.json
  if [ $mode = three_boards ]; then export AZH_NO_THIN_TAIL=1; else unset AZH_NO_THIN_TAIL; fi
  t0=$(date +%s.%N)
  timeout -k 10 400 python3 accelerated_generate_games.py --network /tmp/thin-ab.npy --output-games /tmp/thin-ab-$mode.json --visits 400 \
      --buffer-size ${BUF:-1024} --game-count $N --seed 77 > /tmp/thin-ab-$mode.log 2>&1 || { tail -3 /tmp/thin-ab-$mode.log; exit 1; }
  t1=$(date +%s.%N)
  echo "$mode: $N games in $(python3 -c "print('%.1f' % ($t1-$t0))") s (process start and weight packing included); $(grep Totals /tmp/thin-ab-$mode.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().split('Totals: ')[1]); print('%.1f s in the loop, %d MCTS steps, %d lines' % (d['seconds'], d['steps'], d['written']))")"
done
