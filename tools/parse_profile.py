#!/usr/bin/env python3
"""Condense a tools/profile.sh run (gpurun_out/prof_<tag>/) into the files committed under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg ns)
  profiles/<tag>_pmc.json           per-launch medians of the PMC passes for the render kernel
  profiles/hbm_traffic.json         HBM bytes per launch that bench.py reports as roofline.traffic
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate passes (TCC slots); on gfx950 FETCH_SIZE tallies 128-B requests of 16-B-per-lane loads at 64 B,
so the read side is doubled; WRITE_SIZE is exact for dword-per-lane stores.
usage: tools/parse_profile.py <tag> [scene]"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    scene = sys.argv[2] if len(sys.argv) > 2 else "heightfield"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    med = {}
    meta = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "renderKernel<false" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
        for k, v in acc.items():
            med[k] = {"median": statistics.median(v), "min": min(v), "max": max(v), "launches": len(v)}
    out = {"kernel": "renderKernel<false, false>", "dispatch": meta, "counters": med}
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        rd = med["FETCH_SIZE"]["median"] * 1024.0 * 2.0   # gfx950 correction, see docstring
        wr = med["WRITE_SIZE"]["median"] * 1024.0
        out["hbm_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr,
                                       "read_raw_counter": med["FETCH_SIZE"]["median"] * 1024.0}
        tfile = os.path.join(dst, "hbm_traffic.json")
        cur = json.load(open(tfile)) if os.path.exists(tfile) else {}
        cur[scene] = {"bytes_per_launch": rd + wr, "from": tag + "_pmc.json"}
        json.dump(cur, open(tfile, "w"), indent=1)
    if "TCC_HIT_sum" in med and "TCC_MISS_sum" in med:
        h, m = med["TCC_HIT_sum"]["median"], med["TCC_MISS_sum"]["median"]
        out["l2_hit_rate"] = h / (h + m)
    json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
