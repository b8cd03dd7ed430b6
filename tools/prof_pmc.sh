#!/bin/bash
# tools/prof_pmc.sh <scene> : PMC passes for the trace kernels of one scene (run through gpurun)
S=${1:-man}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$S; mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 tools/prof_one.py $S > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
run a SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES
run c TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
python3 - <<PY
import csv, glob, collections
for n in 'abcd':
    fs=glob.glob('$OUT/%s/*/*_counter_collection.csv'%n)
    if not fs: print('no output for pass',n); continue
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_trace' in k: d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in d:
        for c,v in d[k].items(): print(n,k,c,'%.4g'%(sum(v)/len(v)))
PY
