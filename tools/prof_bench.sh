#!/bin/bash
# rocprofv3 passes over bench.py itself (run on the GPU box from the repo root):
# kernel trace + stats of the default bench command, then PMC passes for the dominant kernel.
#   ARGS="..."      the bench command's arguments (default: the headline, 6 steps)
#   KERNELS="a b"   kernels to summarise, substrings of their names (default: k_tower k_tree — since round 6 the queued moves are
#                   played inside the tower launch; AZH_REROOT_SIDE_STREAM=1 brings k_advance_list back); the first one's
#                   summary names the sha256 of net_kernels.hip, the others' that of engine.hip
#   NO_TIMED=1      no "last steps x 250 launches" block (for commands whose kernels of interest are a leg's, not the headline's)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${OUT:-$R/gpurun_out/prof_bench}
ARGS=${ARGS:---steps 6 --warmup 2 --no-cpu-baseline --no-target-leg --no-gemm-ceiling}
KERNELS=${KERNELS:-k_tower k_tree}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py $ARGS > $OUT/pmc1.log 2>&1 || exit 2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py $ARGS > $OUT/pmc2.log 2>&1 || exit 3
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- python3 $R/bench.py $ARGS > $OUT/pmc3.log 2>&1 || exit 4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc4 -- python3 $R/bench.py $ARGS > $OUT/pmc4.log 2>&1 || exit 5
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc5 -- python3 $R/bench.py $ARGS > $OUT/pmc5.log 2>&1 || exit 6
# keep the merge small: the per-dispatch counter CSVs are large; summarise on the box
# timed-region average of the tower from the kernel trace (the stats CSV averages over set-up and warm-up too):
# the last steps x 250 launches are the timed ones
python3 - "$OUT" "$ARGS" > $OUT/timed_region.txt <<'PY'
import csv, glob, os, re, sys
f = max(glob.glob(os.path.join(sys.argv[1], "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
m = re.search(r"--steps (\d+)", sys.argv[2])
h = re.search(r"--streams (\d+)", sys.argv[2])
n = (int(m.group(1)) if m else 40) * 250 * (int(h.group(1)) if h else 2)   # bench.py's default: two half-batches in flight
for name in ("k_tower", "k_tree", "k_advance_list"):   # (k_advance_list: side-stream mode only)
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
    d = d[-n:]
    if not d:
        continue
    print("rocprofv3 kernel trace, last %d launches (the timed region): %-10s mean %.1f us  p50 %.1f us" % (
        len(d), name, sum(d) / len(d) / 1e3, sorted(d)[len(d) // 2] / 1e3))
PY
# every summary names the kernel source it measured: bench.py reports roofline.traffic from a committed summary only while
# the source it names is the source in the tree
first=1
for k in $KERNELS; do
  echo "bench.py $ARGS" > $OUT/summary_$k.txt
  if [ $first = 1 ]; then src=net_kernels.hip; else src=engine.hip; fi
  echo "kernel source sha256: $(sha256sum $R/ataxxzero_amd/csrc/$src | cut -d' ' -f1)  (ataxxzero_amd/csrc/$src)" >> $OUT/summary_$k.txt
  if [ $first = 1 ] && [ -z "$NO_TIMED" ]; then cat $OUT/timed_region.txt >> $OUT/summary_$k.txt; fi
  python3 $R/tools/prof_summary.py $OUT $k >> $OUT/summary_$k.txt 2>&1
  first=0
done
cp $(ls $OUT/trace/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats.csv
tail -2 $OUT/trace.log | cut -c1-6000 > $OUT/bench_line_under_the_profiler.txt
find $OUT -name "*_counter_collection.csv" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls -la $OUT
