#!/bin/bash
# development probe: the new tests one group at a time, then the bench with its progress on stderr; every step under its own limit,
# and a step that is killed ends the call
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {
  name=$1; limit=$2; shift 2
  echo "== $name" | tee -a gpurun_out/probe.log
  timeout -k 10 $limit "$@" >> gpurun_out/probe_$name.log 2>&1
  rc=$?
  tail -4 gpurun_out/probe_$name.log
  echo "== $name rc $rc" | tee -a gpurun_out/probe.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
}
: > gpurun_out/probe.log
step tests_new 600 python -m pytest tests -m gpu -x -q --timeout 200 -k "planes_of_every_axis or moving_camera or reference_frame_loop or on_a_box_plane"
step bench 420 python bench.py --steps 50 --no-pmc --no-cpu-baseline
step refcl 500 python -m pytest tests/test_gpu_reference_cl.py tests/test_gpu_tlg.py -m gpu -x -q --timeout 300
step dist 300 python tools/ref_cl_distances.py
