#!/bin/bash
# A/B of tower build variants on ONE device in ONE call (boxes differ by several percent): builds variant libraries
# next to the product library and times the net-only bench with each, interleaved, three rounds.
# usage (GPU box, repo root): bash tools/tower_variants.sh "-DAZH_SETPRIO=1" "-DAZH_SETPRIO=3" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
C=ataxxzero_amd/csrc
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude"
# the other translation units once (csrc/_obj does not travel to the GPU box)
O=gpurun_out/variant_obj
mkdir -p $O
hipcc $COMMON -ffp-contract=off -c $C/engine.hip -o $O/engine.o &
hipcc $COMMON -ffp-contract=off -c $C/rules_api.hip -o $O/rules_api.o &
hipcc $COMMON -c $C/common.cpp -o $O/common.o &
hipcc $COMMON -c $C/json.cpp -o $O/json.o &
hipcc $COMMON -c $C/ref_abi.cpp -o $O/ref_abi.o &
libs=(ataxxzero_amd/libataxxzero_hip.so)
names=(base)
i=0
for flags in "$@"; do
  i=$((i+1))
  out=gpurun_out/variant_$i
  mkdir -p $out
  hipcc $COMMON $flags -c $C/net_kernels.hip -o $out/net_kernels.o &
  libs+=($out/lib.so); names+=("$flags")
done
wait
for k in $(seq 1 $i); do
  out=gpurun_out/variant_$k
  hipcc -shared -fPIC --offload-arch=gfx950 -o $out/lib.so $out/net_kernels.o $O/engine.o $O/rules_api.o $O/common.o $O/json.o $O/ref_abi.o || exit 2
done
for round in 1 2 3; do
  for k in "${!libs[@]}"; do
    echo "round $round [${names[$k]}]: $(AZH_LIB=$R/${libs[$k]} python tools/net_bench.py ${N:-3600 16384} | tr '\n' ' ')"
  done
done
# optional: run the net tests against the first variant (CHECK=1)
if [ -n "$CHECK" ]; then AZH_LIB=$R/gpurun_out/variant_1/lib.so python -m pytest tests/test_gpu_net.py -q -m gpu 2>&1 | tail -2; fi
