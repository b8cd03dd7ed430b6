mkdir -p gpurun_out/r5i
python __graft_entry__.py smoke > gpurun_out/r5i/smoke.log 2>&1; tail -2 gpurun_out/r5i/smoke.log
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extra-legs --details gpurun_out/r5i/d$i.json > gpurun_out/r5i/l$i.json 2> gpurun_out/r5i/e$i.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r5i/d$i.json'))
print($i, d['value'], d['ms_per_step'], {n:(v['strategy']['kernel'][:4], v['strategy']['budget'], v['trace_ms']) for n,v in d['per_scene'].items()})
PY
done
