#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "reference_scene_layer" > gpurun_out/r2_tests6.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2_tests6.log
