#!/bin/bash
# Build an experimental variant of the library next to the product one, for in-process A/B timing with tools/ab_lib.py:
#   tools/variant_build.sh NAME "-DMACRO=1 ..."   ->  directx-raytracer_amd/libcrt_hip_NAME.so
# The kernels (render_kernels.hip, path_kernels.hip) and crt_api.cpp are recompiled with the extra flags; everything else is shared with the product build.
set -e
NAME=$1; FLAGS=$2; HIPONLY=$3   # optional third argument: flags for hipcc only (e.g. "-mllvm -option")
cd "$(dirname "$0")/../directx-raytracer_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 $FLAGS $HIPONLY -c render_kernels.hip -o build/rk_$NAME.o &
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 $FLAGS $HIPONLY -c path_kernels.hip -o build/pk_$NAME.o &
wait
g++ -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $FLAGS -c crt_api.cpp -o build/api_$NAME.o
g++ -shared -o ../libcrt_hip_$NAME.so build/scene.o build/image_decode.o build/jpeg_decode.o build/scene_parser.o build/bvh_build.o build/api_$NAME.o build/renderer.o build/rk_$NAME.o build/pk_$NAME.o build/bvh_gpu.o \
    -L/opt/rocm/lib -lamdhip64 -ldl -fopenmp -Wl,-rpath,/opt/rocm/lib
rm -f build/rk_$NAME.o build/pk_$NAME.o build/api_$NAME.o
echo "built directx-raytracer_amd/libcrt_hip_$NAME.so"
