#!/bin/bash
# SQ counters of the path-tracing kernel variant (tools/path_bench.py --quick = C3, 1080p, 4 spp, 3 bounces)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_path
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/path_bench.py --quick > $OUT/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pathKernel<false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s %18.1f  (%d launches)" % (k, statistics.median(v), len(v)))
PY
