"""Data parallelism through the MODEL on the GPU: two child ranks share cuda:0 and reduce over gloo
(RCCL needs one device per rank; the phase split, the bucket hand-over, the communication stream and the reducer are the
code the 8-GPU run uses, only the transport differs).  Checks, against the CPU oracle:

* both ranks receive rank 0's weights (state broadcast),
* each rank's `.grad` after backward() is the AVERAGE over ranks of the per-shard gradients (per-replica BatchNorm
  statistics, as the reference's single-device semantics give each replica: no SyncBN) = what the oracle computes shard
  by shard,
* after optimizer.step() both ranks hold bit-identical parameters,
* `python bench.py --gpus 2` with no launcher starts two ranks itself and reports n_gpus = 2.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from _util import ROOT, pkg, rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / max(float(a.norm() * b.norm()), 1e-300))


def test_two_ranks_through_the_model_average_per_shard_oracle_gradients(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    P, synth, dp = pkg(), pkg("synth"), pkg("dp")
    world, per_rank = 2, 2
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    worker = os.path.join(ROOT, "tests", "helpers", "dp_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, str(tmp_path), str(per_rank)], env=env)
             for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world, codes
    recs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]

    # the oracle, shard by shard, from rank 0's weights
    holder = P.QuadtreeCNN(12, dropout_rate=0.0)
    sd0 = synth.synth_state_dict(holder, salt=0)
    keys = o.trainable_keys(sd0, False)
    G = per_rank * world
    x, f, y = synth.synth_images(G, salt=500), synth.synth_pose_features(G, salt=500), synth.synth_labels(G, 12, salt=500)
    avg, losses = None, []
    for r in range(world):
        b, e = dp.shard_range(G, r, world)
        assert tuple(recs[r]["shard"]) == (b, e)
        sd = o.unique_params(sd0, keys)
        loss = torch.nn.functional.cross_entropy(o.quadtree_forward(sd, x[b:e], f[b:e], train=True, dropout_p=0.0), y[b:e])
        loss.backward()
        losses.append(loss.item())
        g = {k: sd[k].grad / world for k in keys}
        avg = g if avg is None else {k: avg[k] + g[k] for k in keys}
    for r in range(world):
        assert abs(recs[r]["loss"] - losses[r]) <= 1e-3 * max(1.0, abs(losses[r])), (r, recs[r]["loss"], losses[r])
        assert recs[r]["bytes_reduced"] == sum(4 * ((v.numel() + 3) // 4 * 4) for v in recs[r]["grads"].values())
    # identical gradients on both ranks = the average of the per-shard oracle gradients
    assert set(recs[0]["grads"]) == set(keys)
    for k in keys:
        assert torch.equal(recs[0]["grads"][k], recs[1]["grads"][k]), k
        got, want = recs[0]["grads"][k], avg[k]
        if k.split(".")[0] in ("classifier", "numerical_mlp", "quadrant_processor"):
            assert rel_err(got, want) <= 1e-4, (k, rel_err(got, want))
        else:   # through ReLU masks: flip-aware bound (tests/test_model_gpu.py)
            assert _cos(got, want) >= 0.999 and rel_err(got, want) <= 6e-2, (k, _cos(got, want), rel_err(got, want))
    # bit-identical parameters after the step, different from the start, rank 1 started from rank 0's weights
    for k, v in recs[0]["params"].items():
        assert torch.equal(v, recs[1]["params"][k]), k
    assert not torch.equal(recs[0]["params"]["classifier.0.weight"], sd0["classifier.0.weight"])
    # BatchNorm statistics stay per replica (different shards -> different running means)
    assert not torch.equal(recs[0]["running_mean"], recs[1]["running_mean"])


@pytest.mark.parametrize("kind,per_rank", [("quadtree3d", 2), ("cnn_lstm", 6)])
def test_two_ranks_clip_and_sequence_models(tmp_path, kind, per_rank):
    """The same through Quadtree3DCNN (no plan behind it: one flat bucket at the end of backward, video3d.py::_ClipFunction)
    and CnnLstm at BASELINE config 5's per-GPU size (6 sequences x 16 frames; "8-GPU DP" in BASELINE.json): both ranks end
    backward with identical gradients = the average of the per-shard oracle gradients, identical parameters after a step.
    attach_data_parallel used to be a silent no-op on the clip models (round-2 review)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    P, synth, dp = pkg(), pkg("synth"), pkg("dp")
    world = 2
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    worker = os.path.join(ROOT, "tests", "helpers", "dp_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, str(tmp_path), str(per_rank), kind], env=env)
             for r in range(world)]
    codes = [p.wait(timeout=900) for p in procs]
    assert codes == [0] * world, codes
    recs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    G = per_rank * world
    y = synth.synth_labels(G, 12, salt=500)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    if kind == "quadtree3d":
        T, HW = 4, 64
        holder = P.Quadtree3DCNN(12, sequence_length=T, dropout_rate=0.0)
        sd0 = synth.synth_state_dict(holder, salt=0)
        x = synth.synth_images(G * T, salt=500, size=HW).view(G, T, 3, HW, HW)
        f = synth.synth_pose_features(G * T, salt=500).view(G, T, 47)
        keys = [k for k, v in sd0.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
        make = lambda: o.clip_params(sd0)
        fwd = lambda sd, xs, fs: o.quadtree3d_forward(sd, xs, fs, train=True, dropout_p=0.0)
    else:
        T = 16
        holder = P.CnnLstm(12, sequence_length=T, dropout_rate=0.0)
        sd0 = o.cnn_lstm_sd_to_base(synth.synth_state_dict(holder, salt=0))
        x = synth.synth_images(G * T, salt=500).view(G, T, 3, 224, 224)
        f = synth.synth_pose_features(G * T, salt=500).view(G, T, 47)
        keys = [k for k in sd0 if k.split(".")[0] in ("numerical_mlp", "lstm", "classifier")]   # frozen backbone
        make = lambda: o.unique_params(sd0, keys)
        fwd = lambda sd, xs, fs: o.cnn_lstm_forward(sd, xs, fs, train=True, dropout_p=0.0)
    avg, losses = None, []
    for r in range(world):
        b, e = dp.shard_range(G, r, world)
        assert tuple(recs[r]["shard"]) == (b, e)
        sd = make()
        loss = torch.nn.functional.cross_entropy(fwd(sd, x[b:e], f[b:e]), y[b:e])
        loss.backward()
        losses.append(loss.item())
        g = {k: sd[k].grad / world for k in keys if sd[k].grad is not None}
        avg = g if avg is None else {k: avg[k] + g[k] for k in g}
    for r in range(world):
        assert abs(recs[r]["loss"] - losses[r]) <= 1e-3 * max(1.0, abs(losses[r])), (r, recs[r]["loss"], losses[r])
        assert recs[r]["bytes_reduced"] > 0
    assert set(recs[0]["grads"]) == set(recs[1]["grads"]) and len(recs[0]["grads"]) > 0
    for k in recs[0]["grads"]:
        assert torch.equal(recs[0]["grads"][k], recs[1]["grads"][k]), k
    checked = 0
    for k, want in avg.items():
        if k not in recs[0]["grads"]:
            continue
        got = recs[0]["grads"][k]
        if kind == "quadtree3d" and k.endswith(".0.bias") and (k.startswith("conv3d_")):
            continue      # a conv bias in front of a train-mode BatchNorm: its gradient is rounding noise around zero
        if kind == "quadtree3d" and k.startswith("conv3d_"):   # through ReLU / max-pool decisions: flip-aware bound
            assert _cos(got, want) >= 0.99, (k, _cos(got, want))
        else:
            assert _cos(got, want) >= 0.9999 and rel_err(got, want) <= 5e-3, (k, _cos(got, want), rel_err(got, want))
        checked += 1
    assert checked >= 8, checked
    for k, v in recs[0]["params"].items():
        assert torch.equal(v, recs[1]["params"][k]), k
    assert not torch.equal(recs[0]["running_mean"], recs[1]["running_mean"])


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` without torchrun: bench.py starts the ranks (fresh child processes) and relays rank 0's
    line; n_gpus must be 2 and the global batch 2 x per-GPU batch.  gloo transport: both ranks on cuda:0."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, QTCNN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--profile-steps", "0"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0 and rec["scaling"] == "weak" and rec["forward"]["value"] > 0
    # the line checks itself: two ranks really reduced the 25,986,028 trainable-and-used f32 gradients (SURVEY.md 8e:
    # 103.9 MB per step) in the four phase buckets, and the join's exposed tail was measured
    dpi = rec["data_parallel"]
    assert dpi["world_size"] == 2 and dpi["backend"] == "gloo" and dpi["rccl"] is False
    assert abs(dpi["gradient_bytes_per_step"] - 25986028 * 4) <= 64 * 4, dpi   # (views padded to 16 bytes)
    assert dpi["buckets_per_step"] == 4 and [b[0] for b in dpi["bucket_order"]] == [1, 2, 4, 8]
    assert dpi["exposed_tail_ms_per_step"] is not None and dpi["exposed_tail_ms_per_step"] >= 0
    # asking for 2 GPUs under a 1-rank launcher is an error, not a silent 1-GPU run
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(cmd, env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
