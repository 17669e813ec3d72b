#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "path or c5 or property or batch" > gpurun_out/r2_tests9.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_tests9.log
timeout -k 10 400 python tools/path_bench.py > gpurun_out/r2_path4.log 2>&1; cat gpurun_out/r2_path4.log
for w in 8 32; do CRT_LIB=directx-raytracer_amd/libcrt_hip_rf$w.so timeout -k 10 300 python tools/path_bench.py > gpurun_out/r2_path4_rf$w.log 2>&1; echo "refill_min $w"; grep "mode 200" gpurun_out/r2_path4_rf$w.log; done
