mkdir -p gpurun_out/r5g
timeout -k 10 300 python tools/host_issue_probe.py > gpurun_out/r5g/host_issue.json 2> gpurun_out/r5g/host_issue.err; cat gpurun_out/r5g/host_issue.json; tail -3 gpurun_out/r5g/host_issue.err
