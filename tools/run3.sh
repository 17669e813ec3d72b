#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=directx-raytracer_amd
timeout -k 10 500 python tools/ab_lib.py $D/libcrt_hip_r1.so $D/libcrt_hip.so $D/libcrt_hip_sort.so $D/libcrt_hip_pfl.so $D/libcrt_hip_pfp.so $D/libcrt_hip_pfb.so $D/libcrt_hip_xv30.so > gpurun_out/r2_ab2.log 2>&1; echo "ab rc=$?"
cat gpurun_out/r2_ab2.log
