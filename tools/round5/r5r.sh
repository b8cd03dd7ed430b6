mkdir -p gpurun_out/r5r
for s in cube clumsydragon man; do
  for env in "" "ROC_ACTIVE_WAIT_TIMEOUT=2000" "HIP_LAUNCH_BLOCKING=0 ROC_ACTIVE_WAIT_TIMEOUT=100000"; do
    echo "[$env]" $(env $env python tools/dyn_profile.py $s --mode 0 --in-flight 1 2>/dev/null | grep "static frame")
  done
done > gpurun_out/r5r/dyn.txt 2>&1
cat gpurun_out/r5r/dyn.txt
