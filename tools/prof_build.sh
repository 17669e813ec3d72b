#!/bin/bash
# Build the CRT_PROF diagnostic variant of the kernels (in-kernel s_memtime stamps around node / leaf phases) as
# directx-raytracer_amd/libcrt_hip_prof.so; run it with tools/prof_run.py. Never shipped, never timed as a product number.
set -e
cd "$(dirname "$0")/../directx-raytracer_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -DCRT_PROF=1 -c render_kernels.hip -o build/rk_prof.o
g++ -shared -o ../libcrt_hip_prof.so build/scene.o build/scene_parser.o build/bvh_build.o build/crt_api.o build/renderer.o build/rk_prof.o -L/opt/rocm/lib -lamdhip64 -fopenmp -Wl,-rpath,/opt/rocm/lib
rm -f build/rk_prof.o
