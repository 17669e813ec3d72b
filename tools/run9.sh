#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "phong or headless or native_rccl or rccl_gather" > gpurun_out/r2_tests5.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r2_tests5.log
