#!/bin/bash
# One GPU-box call of a development round: the GPU test suite, then the bench line.  A step that is killed at its limit ends the
# call (no further GPU step behind a hang); a failing test does not.
#   tools/gpu_round.sh <tag> [pytest -k expression]
set -o pipefail
tag=${1:-dev}
sel=${2:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$sel" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "$sel" > gpurun_out/${tag}_tests.log 2>&1
else
  timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
fi
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed at their limit (rc $rc): stopping"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 100 --details gpurun_out/${tag}_bench_details.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
brc=$?
echo "bench rc $brc"; tail -3 gpurun_out/${tag}_bench.err
python - <<PY
import json
try:
    print(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
    d = json.load(open('gpurun_out/${tag}_bench_details.json'))
    print('value', d['value'], 'ms_per_step', d['ms_per_step'])
    print('moving', json.dumps({k: v for k, v in (d.get('moving_camera') or {}).items() if k != 'per_scene'}))
    for n, v in (d.get('moving_camera') or {}).get('per_scene', {}).items(): print(' ', n, v)
    for n, v in d['per_scene'].items(): print(' ', n, v['trace_ms'], v['first_kernel_ms'], v['wave_per_ray_ms'], v.get('first_frame_ms'), v['strategy'])
    print('config5', d['config5'])
except Exception as e:
    print('no bench line:', e)
PY
exit $(( rc != 0 ? rc : brc ))
