mkdir -p gpurun_out/r5q
timeout -k 10 500 python tools/build_soak.py --builds 2000 --every 40 > gpurun_out/r5q/soak.txt 2>&1; echo "rc $?" >> gpurun_out/r5q/soak.txt; tail -6 gpurun_out/r5q/soak.txt
timeout -k 10 400 python tools/build_soak.py --builds 1500 --every 25 --stress > gpurun_out/r5q/soak_stress.txt 2>&1; echo "rc $?" >> gpurun_out/r5q/soak_stress.txt; tail -4 gpurun_out/r5q/soak_stress.txt
