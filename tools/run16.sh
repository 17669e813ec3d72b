#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 bash tools/pmc_path.sh > gpurun_out/r2_pmc_path2.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r2_pmc_path2.log
