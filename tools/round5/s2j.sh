# session 2 of round 5: the diag twin's counters with a 16-entry and a 64-entry mailbox (how many triangle tests are repeats the mailbox forgot)
set -o pipefail
mkdir -p gpurun_out/s2j
for lib in librtmi_exp_tlgdiag.so librtmi_exp_diag64.so; do
RTMI_LIBRARY=$lib RTMI_TLG_DIAG_PRINT=1 timeout -k 10 200 python tools/tlg_config4.py > gpurun_out/s2j/$lib.json 2> gpurun_out/s2j/$lib.err
echo "== $lib"; grep "tlg diag" gpurun_out/s2j/$lib.err | tail -13 | cut -c1-150
done
