# session 2 of round 5: config 4 under experiment twins of the grid walk (libraries named on the command line; "base" = librtmi.so)
set -o pipefail
OUT=gpurun_out/s2d; mkdir -p $OUT
for lib in "$@"; do
  L=$lib; [ "$lib" = base ] && L=""
  RTMI_LIBRARY=$L timeout -k 10 200 python tools/tlg_config4.py > $OUT/tlg_$lib.json 2> $OUT/tlg_$lib.err && python -c "
import json; d=json.load(open('$OUT/tlg_$lib.json')); print('tlg [$lib]', d['build_ms'], d['trace_ms'], d['trace_plus_shade_ms'], d['verified'])" || tail -3 $OUT/tlg_$lib.err
done
