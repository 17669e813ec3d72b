#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "path or c5 or property or batch or textured" > gpurun_out/r2_tests10.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2_tests10.log
timeout -k 10 400 python tools/path_bench.py > gpurun_out/r2_path5.log 2>&1; cat gpurun_out/r2_path5.log
