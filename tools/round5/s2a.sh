# session 2 of round 5: GPU tests + bench of the restored tree, then the grid walk's wave-uniformity counters (diag twin, -DRTMI_TLG_DIAG)
set -o pipefail
bash tools/gpu_round.sh s2base || exit $?
mkdir -p gpurun_out/s2a
RTMI_LIBRARY=librtmi_exp_tlgdiag.so RTMI_TLG_DIAG_PRINT=1 timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/s2a/diag.json 2> gpurun_out/s2a/diag.err
grep "tlg diag" gpurun_out/s2a/diag.err | tail -14
