set -o pipefail
mkdir -p gpurun_out/r5b
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5b/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5b/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 120 tools/micro/pk_rate > gpurun_out/r5b/pk_rate.txt 2>&1 && cat gpurun_out/r5b/pk_rate.txt
for lib in "" librtmi_exp_tlg_noslp.so; do
  RTMI_LIBRARY=$lib timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/r5b/tlg_$lib.json 2> gpurun_out/r5b/tlg_$lib.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r5b/tlg_$lib.json')); print('tlg [$lib]', d['build_ms'], d['trace_ms'], d['verified'])"
done
for lib in "" librtmi_exp_trace_noslp.so; do
  RTMI_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 200 --no-pmc --no-cpu-baseline --no-extra-legs --details gpurun_out/r5b/bench_$lib.json > gpurun_out/r5b/bench_$lib.line 2> gpurun_out/r5b/bench_$lib.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r5b/bench_$lib.json')); print('bench [$lib]', d['value'], d['ms_per_step'], {n: v['trace_ms'] for n, v in d['per_scene'].items()})"
done
exit $rc
