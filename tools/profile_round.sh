#!/bin/bash
# Collects the rocprofv3 evidence for one round (run on the GPU box through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1) kernel trace + stats of the default bench command, 2..4) PMC passes (separate runs, as the guide prescribes).
set -o pipefail
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$R
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1 || exit 1
echo "sq done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_tcc.log 2>&1 || echo "tcc pass failed"
echo "tcc done"
find $OUT -name "*.csv" | head -30
# config 4 (Two-Level Grid build + trace on the atrium-class scene): kernel stats of its own command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tlg_stats -- python3 tools/tlg_config4.py > $OUT/tlg_config4.log 2>&1 || echo "tlg stats pass failed"
echo "tlg done"
# config 3 (Bunny-class BVH frame, 1920x1080): kernel stats of its own command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config3_stats -- python3 tools/config_report.py config3 > $OUT/config3.log 2>&1 || echo "config3 stats pass failed"
echo "config3 done"
