#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 200 python tools/prof_run.py > gpurun_out/r2_prof_run.log 2>&1; echo "prof_run rc=$?"; cat gpurun_out/r2_prof_run.log
timeout -k 10 200 python tools/timeline.py > gpurun_out/r2_timeline.log 2>&1; echo "timeline rc=$?"; tail -45 gpurun_out/r2_timeline.log
timeout -k 10 900 bash tools/profile.sh r02a > gpurun_out/r2_profile_r02a.log 2>&1; echo "profile rc=$?"; tail -12 gpurun_out/r2_profile_r02a.log
