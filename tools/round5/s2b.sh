# session 2 of round 5: the packet walk of the grid (scalar records where the wave's lanes agree): the grid tests, then config 4
set -o pipefail
mkdir -p gpurun_out/s2b
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "tlg or grid or config4 or configs" > gpurun_out/s2b/tests.log 2>&1; rc=$?
tail -5 gpurun_out/s2b/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit $rc; fi
for i in 1 2; do
timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/s2b/tlg_$i.json 2> gpurun_out/s2b/tlg_$i.err && python -c "
import json; d=json.load(open('gpurun_out/s2b/tlg_$i.json')); print('tlg', d['build_ms'], d['trace_ms'], d['trace_plus_shade_ms'], d['verified'])" || tail -3 gpurun_out/s2b/tlg_$i.err
done
