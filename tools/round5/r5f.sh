set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5f; mkdir -p $OUT
python3 assets/synth/gen_scenes.py > /dev/null || exit 1
apps/bin/scene_pack_tool assets/synth/synth_atrium.txt /tmp/_synth_atrium.bin || exit 1
export RTMI_PACKED=1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 tools/tlg_config4.py > $OUT/pmc_sq.log 2>&1 || echo "pmc failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/pmc_sq2 -- python3 tools/tlg_config4.py > $OUT/pmc_sq2.log 2>&1 || echo "pmc2 failed"
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc_sq','pmc_sq2'):
    f = glob.glob('gpurun_out/r5f/%s/**/*counter_collection.csv' % d, recursive=True)
    if not f: print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_tlg_trace' in k:
            a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for c, (n, v) in sorted(agg.items()): print(d, c, n, v / n)
PY
