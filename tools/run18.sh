#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
CRT_BUILD_TIMING=1 timeout -k 10 400 python tools/build_bench.py > gpurun_out/r2_build1.log 2>&1; echo "rc=$?"; grep -v "^\[build\]" gpurun_out/r2_build1.log | tail -8; grep "^\[build\]" gpurun_out/r2_build1.log | tail -60
