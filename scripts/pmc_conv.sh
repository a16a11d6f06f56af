#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counters of single conv launches (scripts/bench_conv.py <mode>), one PMC pass,
# --kernel-trace only (no other trace domain, as gpurun requires).  Usage: scripts/pmc_conv.sh <outdir> <bench_conv mode>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_conv}
MODE=${2:-pt1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/a" -- python3 $R/scripts/bench_conv.py $MODE > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT/b" -- python3 $R/scripts/bench_conv.py $MODE > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    f = glob.glob(f"{out}/{sub}/*/*_counter_collection.csv")
    if not f:
        print(sub, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "conv_" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} {v / n[(k, c)]:16.0f}  (x{n[(k, c)]})")
PY
