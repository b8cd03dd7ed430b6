mkdir -p gpurun_out/r5p
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "bvh_and_grid" > gpurun_out/r5p/tests.log 2>&1; rc=$?; tail -15 gpurun_out/r5p/tests.log; exit $rc
