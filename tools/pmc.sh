#!/bin/bash
# Collect arbitrary PMC passes for the render kernel: tools/pmc.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...
# (each pass is its own short bench run; counters of one pass must fit the per-block slots, see MI355X_MICROARCH.md)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $OUT/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, statistics, json
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "renderKernel<false" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: statistics.median(v) for k, v in sorted(acc.items())}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items():
    print("%-40s %18.1f" % (k, v))
PY
