set -o pipefail
mkdir -p gpurun_out/r5c
timeout -k 10 300 tools/micro/div_check > gpurun_out/r5c/div_check.txt 2>&1; cat gpurun_out/r5c/div_check.txt
timeout -k 10 120 tools/micro/pk_rate > gpurun_out/r5c/pk_rate.txt 2>&1; grep cndmask gpurun_out/r5c/pk_rate.txt
timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/r5c/tlg.json 2> gpurun_out/r5c/tlg.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r5c/tlg.json')); print('tlg', d['build_ms'], d['trace_ms'], d['verified'])"
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5c/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5c/tests.log
exit $rc
