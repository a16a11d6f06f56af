#!/usr/bin/env python3
"""Register / scratch report of every kernel in libqtcnn_hip.so (CPU only): extracts the gfx950 code objects of the
fat binary (llvm-objdump --offloading, in a temporary directory) and prints their kernel descriptors' notes.
    python scripts/spill_report.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd", "libqtcnn_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
rows = []
with tempfile.TemporaryDirectory() as tmp:
    so = shutil.copy(SO, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    for co in sorted(glob.glob(os.path.join(tmp, "*gfx950*"))):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            def g(key):
                m = re.search(rf"\.{key}:\s+(\S+)", blk)
                return m.group(1) if m else "?"
            rows.append((g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"),
                         g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print(f"{'kernel':104s} vgpr agpr sgpr vspill sspill scratchB  ldsB")
for r in sorted(rows):
    print(f"{r[0][:104]:104s} {r[1]:>4} {r[2]:>4} {r[3]:>4} {r[4]:>6} {r[5]:>6} {r[6]:>8} {r[7]:>5}")
spilled = [r for r in rows if r[4] not in ("0", "?")]
print(f"\n{len(rows)} kernels, {len(spilled)} with vgpr_spill_count > 0" + (":" if spilled else ""))
for r in spilled:
    print("  ", r[0], "spills", r[4], "VGPRs,", r[6], "B of scratch")
