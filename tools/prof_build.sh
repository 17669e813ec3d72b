#!/bin/bash
# Build the CRT_PROF diagnostic variant of the kernels (in-kernel s_memtime stamps around node / leaf phases) as
# directx-raytracer_amd/libcrt_hip_prof.so; run it with tools/prof_run.py. Never shipped, never timed as a product number.
exec "$(dirname "$0")/variant_build.sh" prof "-DCRT_PROF=1"
