#!/bin/bash
# rocprofv3 passes over the net-only bench (run on the GPU box from the repo root).
# Separate passes: kernel trace, then PMC groups (never combined with trace domains).
set -o pipefail
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof_net
N=${N:-16384}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/net_bench.py $N > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $R/tools/net_bench.py $N > $OUT/pmc1.log 2>&1 || exit 2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc2 -- python3 $R/tools/net_bench.py $N > $OUT/pmc2.log 2>&1 || exit 3
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc3 -- python3 $R/tools/net_bench.py $N > $OUT/pmc3.log 2>&1 || exit 4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc4 -- python3 $R/tools/net_bench.py $N > $OUT/pmc4.log 2>&1 || exit 5
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc5 -- python3 $R/tools/net_bench.py $N > $OUT/pmc5.log 2>&1 || exit 6
find $OUT -name "*.csv" | head -40
