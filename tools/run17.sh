#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "knobs or tile or frames_in_flight or batch or heightfield_1m or c3_tiled or ragged" > gpurun_out/r2_tests11.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2_tests11.log
timeout -k 10 300 python tools/sweep.py split_units 0 64 128 256 512 1024 2048 > gpurun_out/r2_split.log 2>&1; cat gpurun_out/r2_split.log
timeout -k 10 300 python tools/sweep.py split_units 0 256 1024 --mode 3 > gpurun_out/r2_split_m3.log 2>&1; cat gpurun_out/r2_split_m3.log
timeout -k 10 300 python tools/dist_overhead.py > gpurun_out/r2_dist2.log 2>&1; grep "N=" gpurun_out/r2_dist2.log
