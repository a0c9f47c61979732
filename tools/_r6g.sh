O=gpurun_out/r6g; mkdir -p $O
C="--no-cpu-baseline --no-target-leg --no-gemm-ceiling"
run() { # name, args, kernels, no_timed
  echo "== prof $1" >> $O/progress.log
  OUT=$PWD/gpurun_out/prof_$1 ARGS="$2" KERNELS="$3" NO_TIMED="$4" timeout -k 10 700 bash tools/prof_bench.sh > $O/prof_$1.log 2>&1; echo "prof $1 rc=$?" >> $O/progress.log
  mkdir -p $O/$1; cp gpurun_out/prof_$1/summary_*.txt gpurun_out/prof_$1/kernel_stats.csv gpurun_out/prof_$1/bench_line_under_the_profiler.txt $O/$1/ 2>/dev/null
  rm -rf gpurun_out/prof_$1
}
echo start > $O/progress.log
run default_two_halves "--steps 6 --warmup 2 $C" "k_tower k_tree" ""
run default_one_batch "--streams 1 --steps 6 --warmup 2 $C" "k_tower k_tree" ""
run config4 "--visits 800 --blocks 8 --dtype f16 --steps 6 --warmup 2 $C" "k_tower k_tree" ""
run config2 "--visits 200 --steps 6 --warmup 2 $C" "k_tower k_tree" ""
run games16384_two_halves "--games 16384 --steps 4 --warmup 1 $C" "k_tower k_tree" ""
run config5 "--games 64 --visits 8 --blocks 1 --steps 1 --warmup 0 --phase-fill 0 --iters-per-step 10 --no-cpu-baseline --no-gemm-ceiling --legs config5_arena" "k_tower2_pair k_tree" "1"
echo "== full gpu suite" >> $O/progress.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" >> $O/progress.log
cat $O/progress.log; tail -18 $O/gpu_tests.log
