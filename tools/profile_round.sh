#!/bin/bash
# Collects the rocprofv3 evidence for one round (run on the GPU box through gpurun from the repo root):
#   tools/profile_round.sh r04
# 1) the bench line of the default command (it runs its own rocprofv3 counter passes: FETCH_SIZE, WRITE_SIZE, SQ_* + GRBM; raw rows kept),
# 2) kernel trace + stats of the same bench command (without the nested counter passes), 3) config 4 and config 3 with their own stats and
# counter passes, 4) the ray-buffer entry (kernel stats + counters), 5) construct() per kernel, 6) the host floor of the tiled path.
# Every profiled program stands directly behind `--` and starts no process of its own: the scenes are packed ONCE, up front
# (RTMI_PACKED=1 makes the tools refuse to spawn the packer; a failed packing aborts the round).
set -o pipefail
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$R
mkdir -p $OUT
python3 assets/synth/gen_scenes.py > /dev/null || exit 1
for s in synth_bunny synth_atrium; do apps/bin/scene_pack_tool assets/synth/$s.txt /tmp/_$s.bin || { echo "packing $s failed"; exit 1; }; done
for s in cube gardenlight tenryuu man clumsydragon; do apps/bin/scene_pack_tool assets/scenes/$s.txt /tmp/_$s.bin || { echo "packing $s failed"; exit 1; }; done
export RTMI_PACKED=1
python3 bench.py --keep-pmc $OUT/pmc --details $OUT/bench_details.json > $OUT/bench_line.json 2> $OUT/bench_line.err || exit 1
echo "bench line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tlg_stats -- python3 tools/tlg_config4.py > $OUT/tlg_config4.log 2>&1 || echo "tlg stats pass failed"
echo "tlg done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config3_stats -- python3 tools/config_report.py config3 > $OUT/config3.log 2>&1 || echo "config3 stats pass failed"
echo "config3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raybuffer_stats -- python3 tools/rays_profile.py > $OUT/raybuffer.log 2>&1 || echo "ray-buffer stats pass failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/build_stats -- python3 tools/build_profile.py clumsydragon synth_atrium --k 20 > $OUT/build.log 2>&1 || echo "build stats pass failed"
# construct() as a chain of launches: the last replayed build of a kernel trace, stage by stage against SURVEY.md 8(d)'s build-stage bytes
for m in 0 2; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/build_tl_dragon_$m -- python3 tools/build_profile.py clumsydragon --k 5 --mode $m > $OUT/build_tl_dragon_$m.log 2>&1 && python3 tools/build_timeline.py $OUT/build_tl_dragon_$m --n 50032 --json $OUT/build_timeline_dragon_mode$m.json > $OUT/build_timeline_dragon_mode$m.txt || echo "build timeline (dragon, mode $m) failed"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/build_tl_atrium_$m -- python3 tools/build_profile.py synth_atrium --k 5 --mode $m > $OUT/build_tl_atrium_$m.log 2>&1 && python3 tools/build_timeline.py $OUT/build_tl_atrium_$m --n 262656 --json $OUT/build_timeline_atrium_mode$m.json > $OUT/build_timeline_atrium_mode$m.txt || echo "build timeline (atrium, mode $m) failed"
done
# ... and host-timed, without a profiler: every bundled scene, the stand-ins and the 1 M-triangle height field, FIXED and FAST; the dynamic-scene frame
python3 tools/build_profile.py cube gardenlight tenryuu man clumsydragon synth_bunny synth_atrium --k 50 --out $OUT/build_times_fixed.json > /dev/null 2> $OUT/build_times.err || echo "build times (FIXED) failed"
python3 tools/build_profile.py cube gardenlight tenryuu man clumsydragon synth_bunny synth_atrium --k 50 --mode 2 --out $OUT/build_times_fast.json > /dev/null 2>> $OUT/build_times.err || echo "build times (FAST) failed"
for s in cube gardenlight tenryuu man clumsydragon; do for m in 0 2; do python3 tools/dyn_profile.py $s --mode $m --in-flight 5 2>/dev/null | grep "static frame"; done; done > $OUT/dynamic_frames.txt
echo "ray buffer + build done"
# counter passes (BASELINE configs[2] asks for the HBM-bandwidth run of the Bunny-class frame): one pass per counter group
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/config3_pmc_$tag -- python3 tools/config_report.py config3 > $OUT/config3_pmc_$tag.log 2>&1 || echo "config3 counter pass $tag failed"
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/tlg_pmc_$tag -- python3 tools/tlg_config4.py > $OUT/tlg_pmc_$tag.log 2>&1 || echo "config4 counter pass $tag failed"
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/raybuffer_pmc_$tag -- python3 tools/rays_profile.py > $OUT/raybuffer_pmc_$tag.log 2>&1 || echo "ray-buffer counter pass $tag failed"
done
# the grid walk's instruction mix: counts per class, to be priced with the per-class issue costs of tools/micro/pk_rate.hip (profiles/r05_valu_issue_costs.txt)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $OUT/tlg_pmc_mix -- python3 tools/tlg_config4.py > $OUT/tlg_pmc_mix.log 2>&1 || echo "config4 instruction-mix pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/tlg_pmc_cyc -- python3 tools/tlg_config4.py > $OUT/tlg_pmc_cyc.log 2>&1 || echo "config4 VALU-cycles pass failed"
# ... and every instruction it issues, by kind (the launch time follows their total: profiles/r05_tlg_instruction_counts.json)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/tlg_pmc_kinds -- python3 tools/tlg_config4.py > $OUT/tlg_pmc_kinds.log 2>&1 || echo "config4 instruction-kinds pass failed"
echo "config counter passes done"
tools/micro/pk_rate > $OUT/pk_rate.txt 2>&1 || echo "pk_rate failed"
# (the exhaustive comparison of the short division sequences takes minutes and changes only with the sequences: run when the binary was built for it,
#  hipcc --offload-arch=gfx950 -O3 tools/micro/div_check.hip -o tools/micro/div_check; the record in profiles/ is kept otherwise)
if [ -x tools/micro/div_check ]; then tools/micro/div_check > $OUT/div_check.txt 2>&1 || echo "div_check failed"; fi
unset RTMI_PACKED
python3 tools/build_profile.py heightfield --k 10 --out $OUT/build_times_heightfield_fixed.json > /dev/null 2>> $OUT/build_times.err; python3 tools/build_profile.py heightfield --k 10 --mode 2 --out $OUT/build_times_heightfield_fast.json > /dev/null 2>> $OUT/build_times.err
python3 tools/config_report.py > $OUT/config_report.json 2> $OUT/config_report.err || echo "config report failed"
python3 tools/rays_sweep.py --out $OUT/raybuffer_sweep.json > $OUT/raybuffer_sweep.log 2>&1 || echo "ray-buffer sweep failed"
RTMI_BENCH_GROUP=4 python3 bench.py --steps 100 --warmup 5 --no-extra-legs --details $OUT/bench_group4.json > $OUT/bench_group4_line.json 2> $OUT/bench_group4.err || echo "group run failed"
echo "group done"
find $OUT -name "*.csv" | head -40
