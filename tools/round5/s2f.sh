# session 2 of round 5: config 4 timing (3 x), then scalar-cache counters of the grid walk
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-s2f}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "tlg or grid or config4" > $OUT/tests.log 2>&1; rc=$?
tail -2 $OUT/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc"; exit $rc; fi
for i in 1 2 3; do
timeout -k 10 200 python tools/tlg_config4.py > $OUT/tlg_$i.json 2> $OUT/tlg_$i.err && python -c "
import json; d=json.load(open('$OUT/tlg_$i.json')); print('tlg', d['build_ms'], d['trace_ms'], d['trace_plus_shade_ms'], d['verified'])" || tail -3 $OUT/tlg_$i.err
done
python3 assets/synth/gen_scenes.py > /dev/null || exit 1
apps/bin/scene_pack_tool assets/synth/synth_atrium.txt /tmp/_synth_atrium.bin || exit 1
export RTMI_PACKED=1
timeout -k 10 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/pmc_sqc -- python3 tools/tlg_config4.py > $OUT/pmc_sqc.log 2>&1 || echo "pmc sqc failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d $OUT/pmc_sq -- python3 tools/tlg_config4.py > $OUT/pmc_sq.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in ('pmc_sqc','pmc_sq'):
    f = glob.glob('%s/%s/**/*counter_collection.csv' % (out, d), recursive=True)
    if not f: print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_tlg_trace<' in k:
            a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for c, (n, v) in sorted(agg.items()): print(d, c, n, v / n)
PY
