# session 2 of round 5: the grid tests (with the fused frame and the moving cameras), config 4, the bench line
set -o pipefail
OUT=gpurun_out/s2o; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "tlg or grid or config4 or bench" > $OUT/tests.log 2>&1; rc=$?
tail -4 $OUT/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit $rc; fi
timeout -k 10 200 python tools/tlg_config4.py > $OUT/tlg.json 2> $OUT/tlg.err && python -c "
import json; d=json.load(open('$OUT/tlg.json')); print('tlg', d['build_ms'], d['trace_ms'], d['trace_plus_shade_ms'], d['trace_then_shade_two_launches_ms'], d['mrays_per_s'], d['verified'])" || tail -3 $OUT/tlg.err
timeout -k 10 400 python bench.py --details $OUT/bench_details.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['legs']['config4']), d['verified'].get('config4'))
PY
