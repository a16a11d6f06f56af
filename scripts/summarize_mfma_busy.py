#!/usr/bin/env python3
"""gpurun_out/<name>/{train,eval} (scripts/collect_mfma_busy.sh) -> profiles/<tag>_mfma_busy.json.

SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the matrix pipes over all SIMDs of the chip: a
v_mfma_f32_16x16x32_bf16 (16,384 FLOP) holds its pipe for 16 cycles, so busy / 16 is the number of MFMAs executed
and busy / (1024 SIMDs x elapsed cycles) the utilisation.  Elapsed time is NOT taken from the profiled run (counters
serialise kernels and lower the clock): pass the un-profiled ms per step of the same command."""
import collections
import csv
import glob
import json
import os
import sys

name, tag = sys.argv[1], sys.argv[2]
ms = {"train": float(sys.argv[3]), "eval": float(sys.argv[4])}
steps = 5  # bench.py --steps 3 --warmup 2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
algorithmic = {"train": 11.0792e9 * 256, "eval": 3.7718e9 * 256}
FAMS = ("conv_igemm", "conv_pt", "conv_s2", "stem_wgrad_rows", "conv_wgrad_tile", "conv_wgrad_patch", "conv_wgrad_s2", "conv_wgrad", "conv_l1_ring", "conv_stem", "linear_splitk")
sys.path.insert(0, root)
import bench  # noqa: E402
out = {"kind": "mfma_busy", "model": "quadtree", "kernel_sources_sha1": bench.kernel_sources_sha1(), "counter": "SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc, one pass, --kernel-trace only)", "batch": 256, "dtype": "bf16",
       "simds": 1024, "clock_ghz_for_utilisation": 2.4, "busy_cycles_per_mfma_16x16x32": 16}
for mode in ("train", "eval"):
    f = max(glob.glob(os.path.join(root, "gpurun_out", name, mode, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # (the latest run: gpurun merges every call into the same directory)
    busy = collections.defaultdict(float)
    launches = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        fam = next((k for k in FAMS if k in r["Kernel_Name"]), "other")
        busy[fam] += float(r["Counter_Value"])
        launches[fam] += 1
    total = sum(busy.values()) / steps
    executed_flop = total / 16 * 16384
    out[mode] = {
        "mfma_busy_cycles_per_step": int(total),
        "executed_mfma_flop_per_step": executed_flop,
        "algorithmic_flop_per_step": algorithmic[mode],
        "executed_over_algorithmic": round(executed_flop / algorithmic[mode], 4),
        "ms_per_step_unprofiled": ms[mode],
        "mfma_utilisation": round(total / (1024 * ms[mode] * 1e-3 * 2.4e9), 4),
        "by_family": {k: {"launches_per_step": launches[k] / steps, "busy_cycles_per_step": int(v / steps)}
                      for k, v in busy.items() if v > 0},
    }
path = os.path.join(root, "profiles", f"{tag}_mfma_busy.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({m: {k: out[m][k] for k in ("executed_over_algorithmic", "mfma_utilisation")} for m in ("train", "eval")}))
