#!/usr/bin/env python3
"""Condense a tools/profile.sh run (gpurun_out/prof_<tag>/) into the files committed under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg ns)
  profiles/<tag>_pmc.json           per-launch medians of every PMC pass for the render kernel
  profiles/roofline_inputs.json     per scene: what bench.py's roofline block reads (instruction counts, HBM / L2 bytes per launch)
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate passes (TCC slots); on gfx950 FETCH_SIZE tallies 128-B requests of 16-B-per-lane loads at 64 B,
so the read side is doubled; WRITE_SIZE is exact for dword-per-lane stores.
usage: tools/parse_profile.py <tag> [config] [scene]      config: c3 (default) | c5"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    config = sys.argv[2] if len(sys.argv) > 2 else "c3"
    scene = sys.argv[3] if len(sys.argv) > 3 else None
    key = config if not scene else "%s:%s" % (config, scene)
    kernel = "pathKernel<false" if config == "c5" else "renderKernel<false, false"
    sys.path.insert(0, ROOT)
    import bench  # kernel_source_hash(): the sources these counters belong to
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
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
        for k, v in acc.items():
            med[k] = {"median": statistics.median(v), "min": min(v), "max": max(v), "launches": len(v)}
    out = {"kernel": kernel + "...>", "workload": key, "dispatch": meta, "counters": med}
    m = lambda k: med[k]["median"] if k in med else None
    # the hash written on the GPU box beside the counters; a profile directory from before that was recorded is taken to belong to
    # the working tree (parse it before editing the kernels)
    hash_file = os.path.join(src, "kernel_source_hash.txt")
    src_hash = open(hash_file).read().strip() if os.path.exists(hash_file) else bench.kernel_source_hash()
    inputs = {"from": "profiles/%s_pmc.json" % tag, "kernel_source_hash": src_hash}
    try:
        inputs["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        pass
    if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
        rd = m("FETCH_SIZE") * 1024.0 * 2.0   # gfx950 correction, see docstring
        wr = m("WRITE_SIZE") * 1024.0
        out["hbm_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr, "read_raw_counter": m("FETCH_SIZE") * 1024.0}
        inputs["hbm_read_bytes"], inputs["hbm_write_bytes"] = rd, wr
    if m("TCC_HIT_sum") is not None and m("TCC_MISS_sum") is not None:
        out["l2_hit_rate"] = m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum"))
        inputs["l2_requests"] = m("TCC_HIT_sum") + m("TCC_MISS_sum")
    if m("TCC_REQ_sum") is not None:
        inputs["l2_requests"] = m("TCC_REQ_sum")
    if m("SQ_INSTS_VALU") is not None:
        inputs["valu_insts"] = m("SQ_INSTS_VALU")
    if m("SQ_THREAD_CYCLES_VALU") is not None:
        inputs["thread_cycles_valu"] = m("SQ_THREAD_CYCLES_VALU")
        if m("SQ_INSTS_VALU"):
            out["lanes_active"] = m("SQ_THREAD_CYCLES_VALU") / (m("SQ_INSTS_VALU") * 64.0)
    # dynamic instruction mix -> issue cycles (bench.py roofline_block): the counters name three classes -- fp32 fma / add / mul
    # (full rate), transcendentals, conversions (half rate) -- and lump the rest: 32- and 64-bit integer instructions (add / and /
    # or full rate; min / max / compare / shift / bit-field half) and everything uncounted (float min / max, compares, selects:
    # half rate; moves: full).  bench.py prices the lump at full rate for `frac` (a LOWER bound of the issue cycles) and at half
    # rate for `frac_upper`.
    if all(m(k) is not None for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32")):
        inputs["valu_fp32_fma_add_mul_insts"] = m("SQ_INSTS_VALU_FMA_F32") + m("SQ_INSTS_VALU_ADD_F32") + m("SQ_INSTS_VALU_MUL_F32")
        inputs["valu_cvt_insts"] = m("SQ_INSTS_VALU_CVT") or 0.0
        inputs["valu_trans_insts"] = m("SQ_INSTS_VALU_TRANS_F32") or 0.0
        out["valu_mix"] = {k: m(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                                              "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64")}
    for k in ("TA_BUSY_avr", "TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TA_TOTAL_WAVEFRONTS_sum", "GRBM_GUI_ACTIVE"):
        if m(k) is not None:
            inputs[k.lower()] = m(k)
    for k in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVES"):
        if m(k) is not None:
            inputs[k.lower()] = m(k)
    json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
    rfile = os.path.join(dst, "roofline_inputs.json")
    cur = json.load(open(rfile)) if os.path.exists(rfile) else {}
    cur[key] = inputs
    json.dump(cur, open(rfile, "w"), indent=1)
    print(json.dumps(out, indent=1))
    print(json.dumps(inputs, indent=1))


if __name__ == "__main__":
    main()
