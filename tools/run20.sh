#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "gpu_bvh or lifecycle or textured or state_machine" > gpurun_out/r2_tests13.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2_tests13.log
CRT_BUILD_TIMING=1 timeout -k 10 400 python tools/build_bench.py > gpurun_out/r2_build3.log 2>&1; echo "rc=$?"; grep -v "^\[build\]\|^\[sah" gpurun_out/r2_build3.log | tail -8; grep "^\[build\]" gpurun_out/r2_build3.log | head -10
