#!/bin/bash
# Build the diagnostic variant of the library (options "timeline", "debug_skip_units", "debug_force_measure" compiled in) as
# directx-raytracer_amd/libcrt_hip_diag.so; used by tools/timeline.py and tools/moving_camera.py. Never shipped or timed as a product number.
exec "$(dirname "$0")/variant_build.sh" diag "-DCRT_DIAG=1"
