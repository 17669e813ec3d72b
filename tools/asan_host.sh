#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the CPU test suite: the C++ host code of the library (scene layer, parsers, BVH
# builders, C ABI) is rebuilt with -fsanitize=address,undefined and the tests that need no GPU run against it.
# (GPU code cannot be sanitised on this pool; the HIP kernels are linked in unchanged.)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/directx-raytracer_amd/csrc"
make -j8 > /dev/null
mkdir -p build/asan
for f in scene image_decode jpeg_decode scene_parser bvh_build crt_api renderer; do
  g++ -std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
      -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $f.cpp -o build/asan/$f.o
done
g++ -shared -o ../libcrt_hip_asan.so build/asan/{scene,image_decode,jpeg_decode,scene_parser,bvh_build,crt_api,renderer}.o build/render_kernels.o build/path_kernels.o build/bvh_gpu.o \
    -L/opt/rocm/lib -lamdhip64 -ldl -fopenmp -fsanitize=address,undefined -Wl,-rpath,/opt/rocm/lib
cd "$ROOT"
# libstdc++ is preloaded beside libasan so that the __cxa_throw interceptor resolves inside the python process
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 CRT_HIP_LIBRARY="$ROOT/directx-raytracer_amd/libcrt_hip_asan.so" \
python -m pytest tests -x -q -m "not gpu" -k "not tiling" -p no:cacheprovider
rm -f "$ROOT/directx-raytracer_amd/libcrt_hip_asan.so"
