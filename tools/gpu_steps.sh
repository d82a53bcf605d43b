#!/bin/bash
# Run GPU steps one after another on the GPU box: each under its own `timeout -k 10`, and NO further step once one was killed
# by its timeout (a hung kernel: stop, read, change the code).  An ordinary failure (a red test) does not stop the sequence.
#   tools/gpu_steps.sh "<seconds> <command ...>" "<seconds> <command ...>" ...
mkdir -p gpurun_out
for step in "$@"; do
    secs=${step%% *}; cmd=${step#* }
    echo "[gpu_steps $(date +%H:%M:%S)] (limit ${secs}s) $cmd"
    timeout -k 10 "$secs" bash -c "$cmd"
    rc=$?
    echo "[gpu_steps $(date +%H:%M:%S)] exit $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[gpu_steps] step killed at its limit: stopping here"; exit $rc
    fi
done
exit 0
