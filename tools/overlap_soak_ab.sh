#!/bin/bash
# The generator's round trip at full scale, both orders: tools/soak.sh for $1 seconds (default 200: a game generation at
# 400 visits takes ~70 s, whole 37-KB lines start to flow after that) with the default order — fetch, enqueue the next run,
# then format and write — and with AZH_SEQUENTIAL_DRAIN=1 (run, wait, format, write, run), same seed, one box, one call.
# Compare the `steps` totals of the two Totals lines (same wall time): that ratio is what the overlap buys.
R=${GRAFT_REPO_ROOT:-$PWD}
SECS=${1:-200}
for mode in overlapped sequential; do
  echo "=== $mode ==="
  if [ $mode = sequential ]; then export AZH_SEQUENTIAL_DRAIN=1; else unset AZH_SEQUENTIAL_DRAIN; fi
  EXTRA="--seed 20261004 ${EXTRA_FLAGS}" bash $R/tools/soak.sh $SECS || exit 1
  grep "^Totals" $R/gpurun_out/soak/log.txt
done
