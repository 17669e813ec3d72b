#!/bin/bash
# Profile bench.py on the GPU box: per-kernel time (kernel-trace + stats) and, in separate passes, the HBM counters.
# usage: tools_profile.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_l2.log 2>&1
find $OUT -name "*.csv" | head -40
