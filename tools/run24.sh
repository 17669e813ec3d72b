#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 bash tools/profile.sh r02d_5m heightfield5m > gpurun_out/r2_profile_5m.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r2_profile_5m.log
