mkdir -p gpurun_out/r5m
for rep in 1 2 3; do
for lib in "" librtmi_exp_xnackoff.so; do
  RTMI_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 200 --no-pmc --no-cpu-baseline --no-extra-legs --details gpurun_out/r5m/bench_$lib.json > gpurun_out/r5m/bench_$lib.line 2> gpurun_out/r5m/bench_$lib.err && python -c "
import json; d=json.load(open('gpurun_out/r5m/bench_$lib.json')); print('bench 200 [$lib]', d['value'], d['ms_per_step'], {n: v['trace_ms'] for n, v in d['per_scene'].items()})" || tail -2 gpurun_out/r5m/bench_$lib.err
done
done
for lib in "" librtmi_exp_xnackoff.so; do
  RTMI_LIBRARY=$lib timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/r5m/tlg_$lib.json 2> gpurun_out/r5m/tlg_$lib.err && python -c "
import json; d=json.load(open('gpurun_out/r5m/tlg_$lib.json')); print('tlg [$lib]', d['build_ms'], d['trace_ms'], d['verified'])" || tail -2 gpurun_out/r5m/tlg_$lib.err
done
