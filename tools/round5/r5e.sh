set -o pipefail
mkdir -p gpurun_out/r5e
for lib in "" librtmi_exp_m16w7.so librtmi_exp_m16w8.so; do
RTMI_LIBRARY=$lib timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/r5e/tlg_$lib.json 2> gpurun_out/r5e/tlg_$lib.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r5e/tlg_$lib.json')); print('tlg [$lib]', d['build_ms'], d['trace_ms'], d['verified'])" || tail -3 gpurun_out/r5e/tlg_$lib.err
done
