set -o pipefail
mkdir -p gpurun_out/r5d
timeout -k 10 600 python -m pytest tests/test_gpu_tlg.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5d/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r5d/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 1 0; do
RTMI_TLG_WALK_RECORDS=$v timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/r5d/tlg_$v.json 2> gpurun_out/r5d/tlg_$v.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r5d/tlg_$v.json')); print('tlg records=$v', d['build_ms'], d['trace_ms'], d['verified'])"
done
