#!/bin/bash
# Profile bench.py on the GPU box: per-kernel time (kernel-trace + stats) and, each in its OWN pass (never together with a
# trace domain other than --kernel-trace), the PMC counters the roofline block needs.
# usage: tools/profile.sh <tag> [config] [scene]     config: c3 (default) | c5; scene: heightfield5m | soup (c3 only)
# outputs under gpurun_out/prof_<tag>/ ; then tools/parse_profile.py <tag> [config] [scene]
set -e
TAG=${1:-r03}
CONFIG=${2:-c3}
SCENE=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config $CONFIG --no-cpu-baseline --no-extras"
if [ -n "$SCENE" ]; then B="$B --scene $SCENE"; fi
STEPS=50; PSTEPS=8
if [ "$CONFIG" = "c5" ]; then STEPS=10; PSTEPS=3; fi
# the sources these counters belong to, recorded where and when they are measured (parse_profile.py copies it into roofline_inputs.json)
(cd $ROOT && python3 -c "import bench; print(bench.kernel_source_hash())") > $OUT/kernel_source_hash.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps $STEPS --warmup 5 > $OUT/bench_trace.log 2>&1
echo "trace done"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
            "SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" \
            "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64" \
            "TA_BUSY_avr TA_TA_BUSY_sum" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  # each pass under its own timeout: a profiler that aborts (error 38: a group one pass cannot collect) can leave its finaliser
  # stuck, which must cost seconds, not the rest of the call
  timeout -k 5 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$i -- $B --steps $PSTEPS --warmup 2 > $OUT/bench_pmc_$i.log 2>&1 || echo "pmc pass $i FAILED ($pass): see $OUT/bench_pmc_$i.log"
  echo "pmc pass $i done ($pass)"
done
