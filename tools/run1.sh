#!/bin/bash
# scratch driver for one gpurun call: tests, A/B of library variants, bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2_tests1.log
tail -5 gpurun_out/r2_tests1.log
D=directx-raytracer_amd
timeout -k 10 300 python tools/ab_lib.py $D/libcrt_hip_r1.so $D/libcrt_hip.so $D/libcrt_hip_ns1.so $D/libcrt_hip_ns3.so $D/libcrt_hip_w7.so $D/libcrt_hip_w5.so > gpurun_out/r2_ab1.log 2>&1; echo "ab rc=$?"
cat gpurun_out/r2_ab1.log
timeout -k 10 300 python tools/ab_lib.py $D/libcrt_hip_r1.so $D/libcrt_hip.so --mode 3 > gpurun_out/r2_ab1_m3.log 2>&1; cat gpurun_out/r2_ab1_m3.log
timeout -k 10 300 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; echo "bench rc=$?"; cat gpurun_out/r2_bench1.json
