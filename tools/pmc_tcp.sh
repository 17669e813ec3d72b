#!/bin/bash
# L1 (TA / TCP) counters of the headline kernel, one small group per pass: tools/pmc_tcp.sh [tag] -> gpurun_out/pmc_tcp_<tag>/,
# summary.json (per-launch medians) copied to profiles/<tag>_tcp.json by hand after a look.
# Every pass runs under its own short timeout.  Round 2 lost seven GPU-minutes to one pass: rocprofv3 refused a group of three TA
# stall counters ("rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to
# collect" -- the GROUP did not fit one pass, the counters themselves are fine), aborted (signal 6) inside the first dispatch and
# its finaliser then sat there until the call's limit.  So: at most two counters of a block per pass, and a pass that dies costs
# two minutes at worst.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/pmc_tcp_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2"
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
            "TCP_TCP_LATENCY_sum" "TCP_TCC_READ_REQ_LATENCY_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TAGRAM0_REQ_sum"; do
  i=$((i+1))
  if timeout -k 5 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p$i -- $B > $OUT/log_$i.txt 2>&1; then
    echo "pass $i done ($pass)"
  else
    echo "pass $i FAILED ($pass): $(grep -m1 -o 'error code [0-9]*: [^"]*' $OUT/log_$i.txt)"
  fi
done
python3 - <<PY
import csv, glob, collections, statistics, json
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "renderKernel<false, false" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {"median": statistics.median(v), "launches": len(v)} for k, v in sorted(acc.items())}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items():
    print("%-44s %18.1f  (%d launches)" % (k, v["median"], v["launches"]))
PY
