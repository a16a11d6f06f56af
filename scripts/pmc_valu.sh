#!/bin/bash
# Run ON THE GPU BOX (through gpurun): vector-instruction count beside the matrix-pipe busy cycles of every kernel of a
# bench command (one PMC pass, --kernel-trace only).  Usage: scripts/pmc_valu.sh <outdir> [bench.py arguments...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_valu}
shift || true
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p" -- \
    python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 "$@" > "$OUT/p.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/p/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
print(f"{'kernel':66s} calls   us/call  mfma%  valu/mfma salu/mfma lds/mfma")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    if d["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0: continue
    mf = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 16.0
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    print(f"{k:66s} {n[k]:5d} {cyc / n[k] / 2400:9.1f} {100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):6.1f} "
          f"{(d['SQ_INSTS_VALU'] - mf) / mf:9.2f} {d['SQ_INSTS_SALU'] / mf:9.2f} {d['SQ_INSTS_LDS'] / mf:8.2f}")
PY
