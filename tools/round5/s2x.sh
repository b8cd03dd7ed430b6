# session 2 of round 5: the grid walk's three code paths against each other over random cameras (atrium and two bundled scenes)
set -o pipefail
mkdir -p gpurun_out/s2x
( timeout -k 10 500 python tools/tlg_soak.py --scene synth_atrium --res 960x540 --poses 48 && timeout -k 10 300 python tools/tlg_soak.py --scene tenryuu --res 640x480 --poses 48 && timeout -k 10 300 python tools/tlg_soak.py --scene gardenlight --res 640x480 --poses 48 ) > gpurun_out/s2x/soak.txt 2> gpurun_out/s2x/soak.err; rc=$?
cat gpurun_out/s2x/soak.txt; tail -5 gpurun_out/s2x/soak.err; exit $rc
