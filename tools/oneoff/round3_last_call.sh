set -e
mkdir -p gpurun_out/nets /tmp/lc
python -c "from ataxxzero_amd import model; c,b=model.random_init(12,128,seed=1); model.save_model('/tmp/lc/model-001.npy',c,b)"
timeout -k 10 150 python accelerated_generate_games.py --network /tmp/lc/model-001.npy --output-games /tmp/lc/model-001-0.json --visits 400 --buffer-size 1024 --game-count 2000 > gpurun_out/nets/gen1.log 2>&1
python - <<'PY' > gpurun_out/nets/json_bytes_check.txt
import json
n = bad = 0
small = 0
for line in open('/tmp/lc/model-001-0.json'):
    line = line.rstrip('\n')
    n += 1
    if json.dumps(json.loads(line), sort_keys=True, separators=(',', ':')) != line:
        bad += 1
    small += line.count('e-')
print('game lines written by the device loop: %d; lines whose bytes differ from json.dumps(sort_keys, compact) of their own parse: %d; exponent-form floats seen: %d' % (n, bad, small))
assert bad == 0 and n >= 2000
PY
timeout -k 10 120 python train.py --steps 1000 --games /tmp/lc/model-001-0.json --old-path /tmp/lc/model-001.npy --new-path /tmp/lc/model-002.npy > gpurun_out/nets/train1.log 2>&1
cp /tmp/lc/model-002.npy gpurun_out/nets/
timeout -k 10 150 python -m pytest tests/test_gpu_cli.py tests/test_gpu_arena.py -m gpu -x -q > gpurun_out/json_tests.log 2>&1
tail -3 gpurun_out/json_tests.log
timeout -k 10 150 python accelerated_generate_games.py --network /tmp/lc/model-002.npy --output-games /tmp/lc/model-002-0.json --visits 400 --buffer-size 1024 --game-count 2000 > gpurun_out/nets/gen2.log 2>&1
timeout -k 10 120 python train.py --steps 1000 --games /tmp/lc/model-002-0.json --old-path /tmp/lc/model-002.npy --new-path /tmp/lc/model-003.npy > gpurun_out/nets/train2.log 2>&1
cp /tmp/lc/model-003.npy gpurun_out/nets/
tail -2 gpurun_out/nets/gen2.log
