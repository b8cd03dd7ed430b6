mkdir -p gpurun_out/r5j
for dbl in "" "tenryuu" "tenryuu,clumsydragon" "tenryuu,clumsydragon,man,cube" "cube,gardenlight,tenryuu,man,clumsydragon"; do
  echo "== PROBE_DOUBLE=$dbl"
  PROBE_DOUBLE=$dbl timeout -k 10 200 python tools/tail_probe.py 20 5 2>&1 | grep "long\|right behind the long" | cut -c1-260
done > gpurun_out/r5j/double.txt 2>&1
cat gpurun_out/r5j/double.txt
