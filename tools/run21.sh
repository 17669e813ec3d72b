#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "gpu_bvh" > gpurun_out/r2_tests14.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2_tests14.log
timeout -k 10 400 python tools/build_bench.py > gpurun_out/r2_build4.log 2>&1; echo "rc=$?"; grep -v "^\[build\]\|^\[sah" gpurun_out/r2_build4.log | tail -8
