#!/bin/bash
# The CPU side under ASan + UBSan: build, the CPU test suite on the sanitized binaries, the scene-loader mutation fuzz.
#   tools/san/run.sh [fuzz cases, default 10000]  ->  profiles/r04_sanitizers.log (summary), non-zero on any sanitizer report
set -o pipefail
cd "$(dirname "$0")/../.."
ROOT=$PWD
CASES=${1:-10000}
make -s -C tools/san || exit 1
LIBASAN=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:exitcode=97 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=98
export RTMI_ORACLE_LIB=$ROOT/tools/san/_build/librt_oracle.so RTMI_SCENE_PACK_TOOL=$ROOT/tools/san/_build/scene_pack_tool RTMI_HOST_LIB=$ROOT/tools/san/_build/librtmi_host.so
LOG=${SAN_LOG:-profiles/r04_sanitizers.log}
{
  echo "== sanitizers (ASan + UBSan, gcc $(gcc -dumpversion)) on the CPU side: $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "== pytest -m 'not gpu' with librt_oracle.so, scene_pack_tool and librtmi_host.so from tools/san/_build (python under LD_PRELOAD=libasan)"
} > $LOG
LD_PRELOAD=$LIBASAN python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15 >> $LOG
rc1=${PIPESTATUS[0]}
echo "== pytest exit code $rc1" >> $LOG
echo "== scene-loader mutation fuzz: $CASES cases through tools/san/_build/scene_pack_tool" >> $LOG
python tools/fuzz_scene_loader.py --cases $CASES --tool tools/san/_build/scene_pack_tool 2>&1 | tail -25 >> $LOG
rc2=${PIPESTATUS[0]}
echo "== fuzz exit code $rc2" >> $LOG
cat $LOG
[ $rc1 -eq 0 ] && [ $rc2 -eq 0 ]
