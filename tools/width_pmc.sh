#!/bin/bash
# Counter deltas between the tree layouts (64-byte 4-wide nodes, packed 48-byte 4-wide, packed 80-byte 8-wide): one rocprofv3
# --pmc pass per counter group over tools/width_run.py; per-kernel medians -> gpurun_out/pmc_width/summary.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_width
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU" "TA_BUSY_avr" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/width_run.py 16 > $OUT/log_$i.txt 2>&1 && echo "pass $i done ($pass)" || echo "pass $i FAILED ($pass)"
done
python3 - <<PY
import csv, glob, collections, statistics, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "renderKernel<false, false" in k:
            lay = "packed8" if "LayPacked<8>" in k else ("packed4" if "LayPacked<4>" in k else "nodes64")
            acc[lay][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {lay: {c: statistics.median(v[4:] or v) for c, v in sorted(d.items())} for lay, d in acc.items()}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for lay in ("nodes64", "packed4", "packed8"):
    print(lay, {c: round(v) for c, v in res.get(lay, {}).items()})
PY
