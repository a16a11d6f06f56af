#!/usr/bin/env python3
"""Turn gpurun_out/<name>/{trace,fetch,write} (scripts/collect_profiles.sh) into the committed
summaries under profiles/: per-kernel duration table and per-kernel-family HBM traffic.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and reads
exactly 1/2 of the bytes of wide coalesced streams -> bytes = FETCH_SIZE * 1024 * 2;
WRITE_SIZE * 1024 is exact for 16-byte-per-lane stores and float atomics."""
import collections
import csv
import glob
import json
import os
import re
import sys

name = sys.argv[1] if len(sys.argv) > 1 else "profile"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
# optional: the model the passes were collected on (QT_PROFILE_ARGS="--model quadtree3d" -> `quadtree3d`), its batch and dtype;
# bench.py::pmc_summary matches a summary on (kind, model, batch, dtype, kernel sources)
model = sys.argv[3] if len(sys.argv) > 3 else "quadtree"
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", name)


def newest(paths):
    """gpurun merges every call's outputs into the same local directory: take the file of the LATEST run, never glob()[0]"""
    return max(paths, key=os.path.getmtime)


def family(kernel):
    if model != "quadtree":   # other models: the kernel's own name without template / call arguments
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", kernel)          # mangled: length-prefixed name behind the anonymous namespace
        if m:
            n = int(m.group(1))
            return kernel[m.end():m.end() + n]
        k = re.sub(r"\(anonymous namespace\)::", "", kernel)
        k = re.sub(r"^(void|int)\s+", "", k.strip())
        k = re.split(r"[<(]", k, maxsplit=1)[0].strip()
        return k.split("::")[-1] or "other"
    for key in ("conv_igemm_kernel", "conv_pt_kernel", "conv_s2_kernel", "stem_bn_bwd_apply2x2", "stem_wgrad_rows", "conv_wgrad_tile_kernel", "conv_wgrad_patch_kernel", "conv_wgrad_s2_kernel", "conv_wgrad_kernel", "conv_l1_ring_kernel",
                "conv_stem_kernel", "conv_patch_kernel", "wgrad_partial_sum", "bn_bwd_apply", "bn_bwd_reduce",
                "bn_act", "stem_pool_bwd", "stem_bn_bwd_sums", "stem_pool", "pack_weights_batched",
                "multi_tensor_apply"):
        if key in kernel:
            return key
    return "other"


def pmc(sub, counter):
    f = [newest(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")))]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            a = agg[family(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


stats = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))
rows = list(csv.DictReader(open(stats)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
stem = tag if model == "quadtree" else f"{tag}_{model}"
with open(os.path.join(root, "profiles", f"{stem}_kernel_stats.csv"), "w") as out:
    out.write(open(stats).read())
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
summary = {}
for fam in sorted(set(fetch) | set(write)):
    n = max(fetch[fam][1], write[fam][1], 1)
    rd = fetch[fam][0] * 1024 * 2 / max(fetch[fam][1], 1)
    wr = write[fam][0] * 1024 / max(write[fam][1], 1)
    summary[fam] = {"launches_profiled": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                    "hbm_bytes_per_launch": round(rd + wr)}
dur = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    d = dur[family(r["Name"])]
    d[0] += float(r["TotalDurationNs"])
    d[1] += int(r["Calls"])
for fam, (ns, calls) in dur.items():
    summary.setdefault(fam, {})["avg_us"] = round(ns / calls / 1e3, 2)
    summary[fam]["calls"] = calls
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_sources_sha1: ties the summary to the kernel sources it was collected on)
json.dump({"kind": "traffic", "model": model, "batch": batch, "dtype": dtype,
           "command": "bench.py%s --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 (B=%d, %s, 1 GPU)"
                      % ("" if model == "quadtree" else " --model " + model, batch, dtype),
           "kernel_sources_sha1": bench.kernel_sources_sha1(), "steps_profiled": 5,
           "corrections": "read = FETCH_SIZE*1024*2 (gfx950 half-count of wide coalesced reads), write = WRITE_SIZE*1024",
           "families": summary}, open(os.path.join(root, "profiles", f"{stem}_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
