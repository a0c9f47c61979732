# A/B of the generator's host loop: next run enqueued before the finished games are formatted and written (default) against
# the sequential order (AZH_SEQUENTIAL_DRAIN=1).  50 visits so that games finish within the seconds this call has.
set -e
mkdir -p gpurun_out/overlap /tmp/ov
python -c "from ataxxzero_amd import model; c,b=model.random_init(12,128,seed=1); model.save_model('/tmp/ov/m.npy',c,b)"
AZH_SEQUENTIAL_DRAIN=1 timeout -k 5 45 python accelerated_generate_games.py --network /tmp/ov/m.npy --output-games /tmp/ov/seq-0.json --visits 50 --seed 11 --max-seconds 18 > gpurun_out/overlap/sequential.log 2>&1
timeout -k 5 45 python accelerated_generate_games.py --network /tmp/ov/m.npy --output-games /tmp/ov/ovl-0.json --visits 50 --seed 11 --max-seconds 18 > gpurun_out/overlap/overlapped.log 2>&1
python - <<'PY' > gpurun_out/overlap/summary.txt
import json, re
for name in ("sequential", "overlapped"):
    text = open("gpurun_out/overlap/%s.log" % name).read()
    tot = json.loads([l for l in text.splitlines() if l.startswith("Totals: ")][-1][8:])
    path = "/tmp/ov/%s-0.json" % ("seq" if name == "sequential" else "ovl")
    n = bad = size = 0
    seen = set()
    for line in open(path):
        line = line.rstrip("\n")
        n += 1
        size += len(line)
        e = json.loads(line)
        if json.dumps(e, sort_keys=True, separators=(",", ":")) != line:
            bad += 1
        seen.add(json.dumps(e["moves"]))
    print("%-10s: %.2f s, %d steps (%.3f M/s), %d nn evals, %d games finished, %d written (%d distinct, %d with other bytes than json.dumps), %.1f KB per line, exit Totals ring_overflow %d"
          % (name, tot["seconds"], tot["steps"], tot["steps"] / tot["seconds"] / 1e6, tot["nn_evals"], tot["games"], tot["written"], len(seen), bad, size / max(n, 1) / 1e3, tot["ring_overflow"]))
    assert n == tot["written"] and bad == 0
PY
cat gpurun_out/overlap/summary.txt
