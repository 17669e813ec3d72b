#!/bin/bash
# L1 (TA / TCP) counters of the headline kernel, one small group per pass: tools/pmc_tcp.sh -> gpurun_out/pmc_tcp/
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tcp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  # (a pass with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum / TA_ADDR_STALLED_BY_TD_CYCLES_sum never came back
  #  on this pool -- the run was killed after 7 minutes of silence -- so those are not collected)
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p$i -- $B > $OUT/log_$i.txt 2>&1 || echo "pass $i failed"
  echo "pass $i done ($pass)"
done
