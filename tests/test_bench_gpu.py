"""GPU: the driver's own bench command, in its default form, prints ONE JSON line with every object the contract names.
(Round 3 never ran bench.py without --profile-steps 0 / --no-cpu-baseline on the final tree; the driver did, and it crashed.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

from _util import ROOT

pytestmark = pytest.mark.gpu


def _run(extra):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_the_drivers_default_bench_command_prints_its_line():
    rec = _run([])
    assert rec["metric"].startswith("images/sec fwd+bwd QuadtreeCNN") and rec["unit"] == "images/s"
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["value"] > 0
    assert rec["dtype"] == "bf16" and rec["config"]["per_gpu_batch"] == 256 and rec["vs_baseline"] is None
    roof = rec["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["unit"] == "TFLOP/s"
    assert "traffic" in roof and isinstance(roof["traffic_source"], str)
    cpu = rec["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] == "port" and cpu["sample"]
    assert rec["forward"]["value"] > 0 and 0 < rec["forward"]["model_mfma_util"] < 1
    for key in ("hbm", "mfma_pmc"):   # read from profiles/: a number or a stated reason, never an error
        assert isinstance(rec[key], dict) and "error" not in rec[key]


def test_the_clip_model_line_also_prints():
    rec = _run(["--model", "quadtree3d", "--batch", "16", "--no-cpu-baseline"])
    assert rec["value"] > 0 and rec["roofline"]["frac"] > 0 and "traffic" in rec["roofline"]
