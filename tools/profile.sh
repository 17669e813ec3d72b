#!/bin/bash
# Profile bench.py on the GPU box: per-kernel time (kernel-trace + stats) and, each in its OWN pass (never together with a
# trace domain other than --kernel-trace), the PMC counters the roofline block needs.
# usage: tools/profile.sh <tag> [scene]      outputs under gpurun_out/prof_<tag>/ ; then tools/parse_profile.py <tag> [scene]
set -e
TAG=${1:-r02}
SCENE=${2:-heightfield}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --scene $SCENE --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 50 --warmup 5 > $OUT/bench_trace.log 2>&1
echo "trace done"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
            "SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$i -- $B --steps 8 --warmup 2 > $OUT/bench_pmc_$i.log 2>&1
  echo "pmc pass $i done ($pass)"
done
