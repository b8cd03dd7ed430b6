#!/bin/bash
# What do the waves of k_trace_packet4 wait on?  One rocprofv3 --pmc pass per counter group (the program directly behind `--`),
# ManFigure at 1920x1080, isolated launches (<1, 2>).  A pass that is killed at its limit ends the script (no GPU step behind a
# hang); what was collected so far is summarised by tools/p4_wait_summary.py.
#   tools/p4_wait_pmc.sh [scene] [outdir]
scene=${1:-man}
out=${2:-gpurun_out/p4_wait}
export TMPDIR=/tmp
mkdir -p "$out"
apps/bin/scene_pack_tool assets/scenes/$scene.txt /tmp/p4wait_$scene.bin || exit 1
run_pass() {
  tag=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$tag" -- python3 tools/pmc_scene_child.py /tmp/p4wait_$scene.bin 1920 1080 5 > "$out/$tag.log" 2>&1
  rc=$?
  echo "pass $tag rc $rc" | tee -a "$out/passes.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $tag was killed at its limit: stopping" | tee -a "$out/passes.txt"; exit $rc; fi
}
: > "$out/passes.txt"
run_pass sq     SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run_pass wait   SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY
run_pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU
run_pass dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQC_TC_DATA_READ_REQ
run_pass dreq   SQC_DCACHE_REQ_READ_1 SQC_DCACHE_REQ_READ_2 SQC_DCACHE_REQ_READ_4 SQC_DCACHE_REQ_READ_8 SQC_DCACHE_REQ_READ_16 SQC_DCACHE_INPUT_VALID_READYB
run_pass ifetch SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ
run_pass smemlat SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM_NORM SQ_IFETCH_LEVEL SQ_LEVEL_WAVES
python3 tools/p4_wait_summary.py "$out" > "$out/summary.json" && cat "$out/summary.json" | head -120
