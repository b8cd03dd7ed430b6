mkdir -p gpurun_out/r5k
for rep in 1 2 3; do
for v in "" max-ilp; do
  lib=""; [ -n "$v" ] && lib=librtmi_exp_trace_$v.so
  RTMI_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 200 --no-pmc --no-cpu-baseline --no-extra-legs --details gpurun_out/r5k/bench_$v.json > gpurun_out/r5k/bench_$v.line 2> gpurun_out/r5k/bench_$v.err && python -c "
import json; d=json.load(open('gpurun_out/r5k/bench_$v.json')); print('bench 200 [$v]', d['value'], d['ms_per_step'], {n: v['trace_ms'] for n, v in d['per_scene'].items()})" || tail -2 gpurun_out/r5k/bench_$v.err
  RTMI_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extra-legs --details gpurun_out/r5k/bench_$v.json > gpurun_out/r5k/bench_$v.line 2> gpurun_out/r5k/bench_$v.err && python -c "
import json; d=json.load(open('gpurun_out/r5k/bench_$v.json')); print('bench  20 [$v]', d['value'], d['ms_per_step'])" || tail -2 gpurun_out/r5k/bench_$v.err
done
done
