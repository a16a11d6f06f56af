#!/usr/bin/env python3
"""Headline benchmark: images/sec of one QuadtreeCNN training step (forward +
backward + gradient all-reduce + Adam) at 224x224, batch 256 per GPU, bf16.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is what the reference's hot loop does per batch
(/root/reference/Quadtree_from scratch/Quadtree_train.py:62-66): zero_grad,
model(images, numerical), CrossEntropyLoss, backward, Adam(lr 1e-4, wd 1e-4).step.
Inputs are synthetic and already resident in HBM; weights are the deterministic
synthetic fill.  Rank 0 prints ONE JSON line (contract in the task statement)
with two extra objects:
  roofline      the MFMA implicit-GEMM kernels (dominant kernel family), timed per
                launch with HIP events on their own stream in extra steps right
                after the timed region (event pairs would perturb `value`)
  cpu_baseline  the CPU oracle (torch fp32 restatement, "port") on this box's host
                cores, bounded sample, N=1 only
"""
import os

# The plan overlaps its weight-gradient stream with the main chain, and RCCL brings its own
# streams: with ROCm's default of 4 hardware queues per process the side stream ends up sharing a
# queue with the main stream once RCCL is initialised (measured: 9.85 vs 8.32 ms/step).  Must be
# set before the HIP runtime initialises, i.e. before torch touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# benchmarks run on the deterministic synthetic weights: no ImageNet checkpoint wanted, no warning about it
os.environ.setdefault("QTCNN_RESNET18_WEIGHTS", "none")

import argparse
import ctypes
import importlib
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd"

FWD_BWD_GFLOP_PER_IMAGE = 11.0792   # BASELINE.md section 3 (all parameters trainable)
FWD_GFLOP_PER_IMAGE = 3.7718
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # MI355X_MICROARCH.md (dense)
# kernel families (scripts/summarize_profiles.py) behind roofline.achieved: forward + data-gradient conv launches
ROOFLINE_FAMILIES = ("conv_igemm_kernel", "conv_pt_kernel", "conv_s2_kernel", "conv_l1_ring_kernel", "conv_stem_kernel",
                     "conv_stem_pool_kernel", "linear_splitk_kernel")
ROOFLINE_FAMILIES_3D = ("conv_igemm_kernel", "conv3d_first_kernel", "conv3d_c32_kernel")


def kernel_sources_sha1():
    """Identity of the kernel sources a PMC summary under profiles/ was collected on (scripts/summarize_*.py store it)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, PKG, "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, PKG, "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "qtcnn.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


SUMMARY_KEYS = {   # what a reader below indexes; a summary without them is skipped, never indexed (round 3: KeyError 'families')
    "traffic": ("kernel_sources_sha1", "families"),
    "mfma_busy": ("kernel_sources_sha1", "train", "eval"),
}


def summary_identity(rec, path):
    """(kind, model, batch, dtype) of a committed PMC summary.  Summaries written since round 4 carry all four fields;
    older ones are identified by file name / command string (the only other model ever summarised is quadtree3d)."""
    base = os.path.basename(path)
    kind = rec.get("kind") or ("mfma_busy" if base.endswith("_mfma_busy.json") else
                               "traffic" if base.endswith("_traffic.json") else None)
    model = rec.get("model")
    if model is None:
        m = re.search(r"--model\s+(\w+)", str(rec.get("command", "")))
        model = m.group(1) if m else ("quadtree3d" if "quadtree3d" in base else "quadtree")
    return kind, model, int(rec.get("batch", 256)), rec.get("dtype", "bf16")


def summary_problems(rec, kind):
    """Reasons a summary of `kind` cannot be consumed by this file ([] = usable)."""
    bad = ["missing key %r" % k for k in SUMMARY_KEYS[kind] if k not in rec]
    if kind == "traffic" and isinstance(rec.get("families"), dict):
        for fam, v in rec["families"].items():
            if not isinstance(v, dict) or not all(isinstance(v.get(k), (int, float)) for k in
                                                  ("hbm_bytes_per_launch", "launches_profiled")):
                bad.append("family %r lacks hbm_bytes_per_launch / launches_profiled" % fam)
    elif kind == "traffic":
        bad.append("'families' is not a mapping")
    if kind == "mfma_busy":
        for mode in ("train", "eval"):
            if not isinstance(rec.get(mode), dict) or not all(k in rec[mode] for k in
                                                              ("mfma_busy_cycles_per_step", "executed_over_algorithmic")):
                bad.append("%r lacks mfma_busy_cycles_per_step / executed_over_algorithmic" % mode)
    return bad


def pmc_summary(kind, model="quadtree", batch=256, dtype="bf16"):
    """The committed profiles/*.json PMC summary of `kind` ("traffic" | "mfma_busy") for exactly this model / batch / dtype,
    collected on EXACTLY the kernel sources of this tree and carrying the keys the readers use -- else (None, reason).
    PMC counters cannot be collected inside bench.py (rocprofv3 wraps the process), so the numbers are read from the
    committed summary of the same command, but never from one that predates a kernel change, belongs to another model or
    has another shape.  Candidates are ordered by NAME (rNN_ prefix), newest round first: file times are checkout times.
    Never raises."""
    import glob
    try:
        cur = kernel_sources_sha1()
        skipped = []
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json")), key=os.path.basename, reverse=True):
            try:
                rec = json.load(open(f))
            except Exception:
                continue
            if not isinstance(rec, dict) or "kernel_sources_sha1" not in rec:
                continue
            if summary_identity(rec, f) != (kind, model, batch, dtype) or rec["kernel_sources_sha1"] != cur:
                continue
            bad = summary_problems(rec, kind)
            if bad:
                skipped.append("%s: %s" % (os.path.basename(f), "; ".join(bad[:3])))
                continue
            return rec, os.path.relpath(f, ROOT)
        return None, ("no committed %s summary of %s (batch %d, %s) matches the current kernel sources (sha1 %s): re-collect "
                      "with scripts/collect_profiles.sh%s" % (kind, model, batch, dtype, cur[:12],
                                                              "; skipped " + " | ".join(skipped) if skipped else ""))
    except Exception as e:   # a broken profiles/ directory must never take the measurement down
        return None, "profiles/ unreadable: %r" % (e,)


def guarded(what, fn, *a, **kw):
    """Enrichment of the JSON line from side files / extra steps: the measurement is already taken when these run, so a
    failure becomes {"error": ...} on the line instead of a crash before the line is printed (the rule cpu_baseline has
    followed since round 1)."""
    try:
        return fn(*a, **kw)
    except Exception as e:
        print("bench.py: %s failed: %r" % (what, e), file=sys.stderr)
        return {"error": "%s: %r" % (what, e)}


def roofline_traffic(model, batch, dtype, families):
    """(HBM bytes per launch over the kernel `families` behind roofline.achieved, source string) from the committed
    FETCH_SIZE / WRITE_SIZE summary of the same command; (None, reason) if there is none for this tree."""
    try:
        rec, src = pmc_summary("traffic", model, batch, dtype)
        if rec is None:
            return None, src
        fams = rec["families"]
        sel = [fams[k] for k in families if k in fams and fams[k].get("launches_profiled", 0)]
        n = sum(v["launches_profiled"] for v in sel)
        traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in sel) / n) if n else None
        return traffic, src + " (kernel sources sha1 " + rec["kernel_sources_sha1"][:12] + ")"
    except Exception as e:
        return None, "traffic summary unusable: %r" % (e,)


def step_hbm(model, batch, dtype, s_per_step):
    """Whole-step HBM bytes by counter against THIS run's step time."""
    rec, src = pmc_summary("traffic", model, batch, dtype)
    if rec is None:
        return {"bytes_per_step": None, "note": src}
    steps_prof = float(rec.get("steps_profiled", 5))
    per_step = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in rec["families"].values()) / steps_prof
    return {"bytes_per_step": int(per_step), "achieved_tb_s": round(per_step / s_per_step / 1e12, 2),
            "peak_tb_s": 8.0, "source": src, "kernel_sources_sha1": rec["kernel_sources_sha1"][:12]}


def step_mfma_busy(model, batch, dtype, mode, s_per_step):
    """Matrix-pipe busy cycles by counter against THIS run's step time."""
    rec, src = pmc_summary("mfma_busy", model, batch, dtype)
    if rec is None:
        return {"utilisation": None, "note": src}
    r = rec[mode]
    return {"busy_simd_cycles_per_step": r["mfma_busy_cycles_per_step"],
            "executed_over_algorithmic_flop": r["executed_over_algorithmic"],
            "utilisation": round(r["mfma_busy_cycles_per_step"] / (1024 * s_per_step * 2.4e9), 4),
            "source": src + " (SQ_VALU_MFMA_BUSY_CYCLES)", "kernel_sources_sha1": rec["kernel_sources_sha1"][:12]}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE config 2: 256)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--freeze-backbone", action="store_true",
                    help="--model quadtree: the resnet/ variant of the reference (resnet/models.py:77-78: base_cnn frozen, "
                         "train-mode BatchNorm statistics still updated); a secondary line, 3.9454 GFLOP/image")
    ap.add_argument("--seq-len", type=int, default=16, help="--model cnn_lstm: frames per sequence (BASELINE config 5: 16)")
    ap.add_argument("--model", default="quadtree", choices=["quadtree", "attention", "cnn_lstm", "quadtree3d"],
                    help="quadtree = QuadtreeCNN (BASELINE config 2/3, the headline); attention = AttentionHierarchicalCNN "
                         "(reference models.py:6-101, SURVEY.md 8f rank 2), cnn_lstm = CnnLstm (cnn+lstm/models.py:14-89, "
                         "rank 3; --batch counts FRAMES per GPU), quadtree3d = Quadtree3DCNN (3dcnn/models.py:96-214, rank 4, "
                         "BASELINE config 4: clips of --seq-len 8 frames; --batch counts FRAMES per GPU) as secondary lines")
    ap.add_argument("--forward-only", action="store_true", help="time eval-mode forward instead of the train step")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="Adam(lr 1e-4, wd 1e-4) by the package's FusedAdam kernel (default) or torch.optim.Adam(fused=True)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-leg", action="store_true",
                    help="skip the eval-forward timing after the train step (profiling runs: keeps kernel counts per step exact)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--profile-steps", type=int, default=2)
    return ap.parse_args()


def host_cores():
    """Cores this process may really use: min(CPU affinity, cgroup CPU quota).  On the GPU boxes the affinity mask shows
    every core of the host (256) while the container's share is 16: 256 torch threads on a 16-core quota run ~100x
    slower than 16 (measured: 0.17 vs ~70 images/s), so the quota decides.  No quota visible: the documented share of a
    1-GPU box (16), or the affinity count if smaller.  QTCNN_CPU_THREADS overrides."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    if os.environ.get("QTCNN_CPU_THREADS"):
        return max(1, min(aff, int(os.environ["QTCNN_CPU_THREADS"]))), "QTCNN_CPU_THREADS"
    quota = None
    try:   # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, -(-int(q) // int(per)))
    except Exception:
        pass
    if quota is None:
        try:   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, -(-q // per))
        except Exception:
            pass
    if quota is not None:
        return max(1, min(aff, quota)), f"min(affinity {aff}, cgroup quota {quota})"
    return max(1, min(aff, 16)), f"min(affinity {aff}, 16 = CPU share of a 1-GPU box; no cgroup quota visible)"


def cpu_baseline(args, num_classes=12):
    """The oracle (kind "port") on the host cores: fwd+bwd+Adam on a bounded sample."""
    import oracle.quadtree_oracle as o  # checker only, never the product path
    P = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    cores, cores_source = host_cores()
    torch.set_num_threads(cores)
    if args.model == "cnn_lstm":
        T = args.seq_len
        holder = P.CnnLstm(num_classes, sequence_length=T)
        sd0 = o.cnn_lstm_sd_to_base(synth.synth_state_dict(holder))
        keys = [k for k in sd0 if k.split(".")[0] in ("numerical_mlp", "lstm", "classifier")]  # frozen backbone
        sd = o.unique_params(sd0, keys)
        opt = torch.optim.Adam([sd[k] for k in keys], lr=1e-4)
        S = max(1, args.cpu_batch // T)
        g = torch.Generator().manual_seed(1234)
        x, f = torch.randn(S, T, 3, 224, 224, generator=g), torch.randn(S, T, 47, generator=g)
        y = torch.randint(0, num_classes, (S,), generator=g)
        iters, t0 = 0, time.perf_counter()
        while iters < 2 or (time.perf_counter() - t0 < 8.0 and iters < 20):
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(o.cnn_lstm_forward(sd, x, f, train=True), y).backward()
            opt.step()
            iters += 1
        dt = time.perf_counter() - t0
        return {"value": round(S * T * iters / dt, 2), "unit": "images/s", "cores": cores, "cores_source": cores_source,
                "kind": "port",
                "sample": f"{iters} train steps of {S} sequences x {T} frames, torch {torch.__version__} CPU fp32, "
                          f"oracle/quadtree_oracle.py::cnn_lstm_forward"}
    if args.model == "quadtree3d":
        # BASELINE config 4: the oracle's Quadtree3DCNN (oracle/quadtree_oracle.py::quadtree3d_forward restates
        # /root/reference/3dcnn/models.py:184-214) on clips of T frames of 224x224: fwd+bwd+Adam, then an eval-forward leg
        T = 8 if args.seq_len == 16 else args.seq_len
        holder = P.Quadtree3DCNN(num_classes, sequence_length=T)
        sd = o.clip_params({k: v.clone() for k, v in synth.synth_state_dict(holder).items()})
        params = [v for v in sd.values() if v.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-4)
        S = max(1, args.cpu_batch // T)
        g = torch.Generator().manual_seed(1234)
        x, f = torch.randn(S, T, 3, 224, 224, generator=g), torch.randn(S, T, 47, generator=g)
        y = torch.randint(0, num_classes, (S,), generator=g)

        def step3d():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(o.quadtree3d_forward(sd, x, f, train=True), y).backward()
            opt.step()

        step3d()  # warm-up
        iters, t0 = 0, time.perf_counter()
        while iters < 2 or (time.perf_counter() - t0 < 10.0 and iters < 20):
            step3d()
            iters += 1
        dt = time.perf_counter() - t0
        out = {"value": round(S * T * iters / dt, 2), "unit": "frames/s", "cores": cores, "cores_source": cores_source,
               "kind": "port",
               "sample": f"{iters} train steps (fwd+bwd+Adam) of {S} clips x {T} frames of 224x224, torch {torch.__version__} "
                         f"CPU fp32, oracle/quadtree_oracle.py::quadtree3d_forward"}
        with torch.no_grad():
            o.quadtree3d_forward(sd, x, f)
            k, t1 = 0, time.perf_counter()
            while k < 1 or (time.perf_counter() - t1 < 5.0 and k < 20):
                o.quadtree3d_forward(sd, x, f)
                k += 1
            out["legs"] = [{"what": f"Quadtree3DCNN eval forward, {S} clips x {T} frames", "unit": "frames/s", "iterations": k,
                            "value": round(S * T * k / (time.perf_counter() - t1), 2)}]
        return out
    if args.model == "attention":
        holder = P.AttentionHierarchicalCNN(num_classes)  # parameter tree only (CPU tensors), never called
        sd0 = o.attention_sd_to_base(synth.synth_state_dict(holder))
        keys = [k for k in o.trainable_keys(sd0, False)]
        forward = o.attention_forward
    else:
        holder = P.QuadtreeCNN(num_classes)
        sd0 = synth.synth_state_dict(holder)
        keys = o.trainable_keys(sd0, False)
        forward = o.quadtree_forward
    sd = o.unique_params(sd0, keys)
    params = [sd[k] for k in keys]
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-4)
    B = args.cpu_batch
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, 224, 224, generator=g)
    f = torch.randn(B, 47, generator=g)
    y = torch.randint(0, num_classes, (B,), generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        logits = forward(sd, x, f, train=True)
        torch.nn.functional.cross_entropy(logits, y).backward()
        opt.step()

    step()  # warm-up
    iters, t0 = 0, time.perf_counter()
    while iters < 2 or (time.perf_counter() - t0 < 8.0 and iters < 20):
        step()
        iters += 1
    dt = time.perf_counter() - t0
    out = {"value": round(B * iters / dt, 2), "unit": "images/s", "cores": cores, "cores_source": cores_source, "kind": "port",
           "sample": f"{iters} train steps (fwd+bwd+Adam) of batch {B}, torch {torch.__version__} CPU fp32, "
                     f"oracle/quadtree_oracle.py"}
    if args.model != "quadtree":
        return out
    # BASELINE.md section 4 legs, each a bounded sample (whole baseline ~20-30 s): config 1 = StandardResNetCNN forward
    # bs 1; QuadtreeCNN forward bs 1; QuadtreeCNN forward bs 256 (one pass).  fwd+bwd at bs 256 needs ~13 GB of f32
    # activations and ~15 s per step on 16 cores: the train-step leg above runs at the stated smaller batch instead.
    legs = []
    del opt, params
    for p in sd.values():
        p.grad = None

    def timed(fn, n_images, budget, max_iters):
        fn()
        k, t = 0, time.perf_counter()
        while k < 1 or (time.perf_counter() - t < budget and k < max_iters):
            fn()
            k += 1
        return round(n_images * k / (time.perf_counter() - t), 2), k

    with torch.no_grad():
        std = P.StandardResNetCNN(num_classes)
        sd_std = synth.synth_state_dict(std)
        v, k = timed(lambda: o.standard_resnet_forward(sd_std, x[:1]), 1, 2.0, 50)
        legs.append({"what": "BASELINE config 1: StandardResNetCNN eval forward, bs 1", "value": v, "unit": "images/s",
                     "iterations": k})
        v, k = timed(lambda: forward(sd, x[:1], f[:1]), 1, 2.0, 50)
        legs.append({"what": "QuadtreeCNN eval forward, bs 1", "value": v, "unit": "images/s", "iterations": k})
        gb = torch.Generator().manual_seed(4321)
        xb, fb = torch.randn(256, 3, 224, 224, generator=gb), torch.randn(256, 47, generator=gb)
        forward(sd, xb[:16], fb[:16])
        t = time.perf_counter()
        forward(sd, xb, fb)
        legs.append({"what": "QuadtreeCNN eval forward, bs 256 (one pass)", "unit": "images/s", "iterations": 1,
                     "value": round(256 / (time.perf_counter() - t), 2)})
    out["legs"] = legs
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay rank 0's JSON line.  Runs BEFORE this process makes any HIP call (a process that has
    initialised the GPU must not exec / fork GPU children on this pool); the children are ordinary subprocesses."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank launch failed (exit code {proc.returncode}, "
              f"{'no' if line is None else 'a'} JSON line from rank 0)", file=sys.stderr)
        raise SystemExit(proc.returncode or 1)
    rec = json.loads(line)
    if rec.get("n_gpus") != args.gpus:
        print(f"bench.py: asked for {args.gpus} GPUs, rank 0 reports {rec.get('n_gpus')}", file=sys.stderr)
        raise SystemExit(1)
    print(line, flush=True)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)       # no launcher: become one (nothing has touched the GPU yet)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > ndev and os.environ.get("QTCNN_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"--gpus {world} but only {ndev} GPU(s) visible: RCCL needs one device per rank "
                         "(QTCNN_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    dist = None
    force_dist = os.environ.get("QTCNN_FORCE_DIST") in ("1", "2")  # rehearse the RCCL path with a single rank
    saved_stdout = None
    if world > 1 or force_dist:
        # RCCL prints a version banner on the process's stdout when the first communicator comes up; the contract is ONE
        # JSON line on stdout, so everything until that line goes to stderr (file-descriptor level: the banner is C code)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # (single-rank rehearsal without a launcher: any free port)
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        # nccl == RCCL over xGMI.  QTCNN_DIST_BACKEND=gloo only exists to rehearse the
        # multi-process path with several ranks on ONE GPU (RCCL refuses duplicate devices).
        backend = os.environ.get("QTCNN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus and not force_dist:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    P = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dp = importlib.import_module(PKG + ".dp")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    C, B = 12, args.batch
    if args.model == "quadtree3d" and args.seq_len == 16:
        args.seq_len = 8   # (the default of --seq-len is config 5's 16; config 4 quotes T = 8)
    if args.model in ("cnn_lstm", "quadtree3d") and B % args.seq_len:
        raise SystemExit("--batch (frames) must be a multiple of --seq-len")
    if args.model == "quadtree3d":
        model = P.Quadtree3DCNN(C, sequence_length=args.seq_len, compute_dtype=dt)
    elif args.model == "cnn_lstm":
        model = P.CnnLstm(C, sequence_length=args.seq_len, compute_dtype=dt, max_batch=B)
    elif args.model == "attention":
        model = P.AttentionHierarchicalCNN(C, compute_dtype=dt, max_batch=B)
    else:
        model = P.QuadtreeCNN(C, compute_dtype=dt, max_batch=B, freeze_backbone=args.freeze_backbone)
    model.load_state_dict(synth.synth_state_dict(model))
    model = model.to(dev)
    if world > 1 or force_dist:
        dp.attach_data_parallel(model)
    trainable = [p for p in model.parameters() if p.requires_grad]
    if args.model == "quadtree" and args.freeze_backbone:
        # resnet/train_cnn_model.py:65 hands every parameter to Adam (lr 1e-4, wd 1e-4); the frozen ones have no gradient
        opt = P.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-4, model=model)
    elif args.model == "cnn_lstm":
        opt = P.FusedAdam(trainable, lr=1e-4, model=model)  # cnn+lstm/training.py:93: Adam(lr 1e-4), no weight decay
    elif args.model == "quadtree3d":
        opt = P.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-4)  # multi-tensor Adam kernel (no plan behind this model)
    elif args.optimizer == "fused":
        # the package's Adam: same update rule, run inside the one-launch weight re-packing (csrc/pack.hip)
        opt = P.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-4, model=model)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    crit = torch.nn.CrossEntropyLoss()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    feats = torch.randn(B, 47, device=dev, generator=g)
    labels = torch.randint(0, C, (B,), device=dev, generator=g)
    if args.model in ("cnn_lstm", "quadtree3d"):
        S, T = B // args.seq_len, args.seq_len
        images, feats, labels = images.view(S, T, 3, 224, 224), feats.view(S, T, 47), labels[:S].contiguous()

    if args.forward_only:
        model.eval()

        def step():
            with torch.no_grad():
                return model(images, feats)
    else:
        model.train()

        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(model(images, feats), labels)
            loss.backward()
            opt.step()
            return loss

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    red0 = getattr(model, "_grad_sync", None)
    if red0 is not None and not args.forward_only:
        red0.measure_tail = True   # two events per join: how long the last compute kernel had been done before the all-reduce was
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_issue = time.perf_counter() - t0   # host time to ISSUE the K steps (no sync): close to `elapsed` = host-bound
    fence()
    elapsed = time.perf_counter() - t0
    exposed_tail = None
    if red0 is not None and red0.measure_tail:
        red0.measure_tail = False
        exposed_tail = red0.exposed_tail_ms()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the MFMA kernels: per-launch HIP events in extra steps ----
    roofline = None
    eng = getattr(model, "_engine", None)
    if eng is not None and args.profile_steps > 0:
        L = eng.L
        L.qt_plan_profile_begin.argtypes = [ctypes.c_void_p]
        L.qt_plan_profile_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
        L.qt_plan_profile_begin(eng.handle)
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        fl, ms, ln = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int * 3)()
        L.qt_plan_profile_end(eng.handle, fl, ms, ln)
        by = (ctypes.c_double * 3)()
        L.qt_plan_profile_bytes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.qt_plan_profile_bytes(eng.handle, by)
        # what the plan's three timing kinds aggregate (csrc/plan.hip begin_timed): every forward conv / linear launch,
        # every data-gradient launch, every weight-gradient launch -- whichever kernel family serves the layer
        kinds = ["forward conv launches (conv_stem / conv_l1_ring / conv_s2 / conv_pt / conv_igemm / linear_splitk)",
                 "data-gradient conv launches (conv_l1_ring / conv_pt / conv_igemm merged stride-2 / linear_splitk)",
                 "weight-gradient launches (conv_wgrad_tile + partial sums / conv_wgrad generic)"]
        per = {}
        for k in range(3):
            if ln[k]:
                per[kinds[k]] = {"launches_per_step": ln[k] // args.profile_steps,
                                 "avg_us": round(1e3 * ms[k] / ln[k], 2),
                                 "gflop_per_launch": round(fl[k] / ln[k] / 1e9, 3),
                                 "tflops": round(fl[k] / (ms[k] * 1e-3) / 1e12, 1),
                                 "algorithmic_mb_per_launch": round(by[k] / ln[k] / 1e6, 1)}
        nig = ln[0] + ln[1]
        if nig:
            ach = (fl[0] + fl[1]) / ((ms[0] + ms[1]) * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            traffic, traffic_src = roofline_traffic(args.model, args.batch, args.dtype, ROOFLINE_FAMILIES)
            roofline = {"kernel": "forward + data-gradient conv launches (conv_pt_kernel for the 3x3 stride-1 layers of "
                                  "layer2-4; conv_l1_ring_kernel for the 56x56 64->64 layers; conv_s2_kernel for the stride-2 "
                                  "transitions forward, conv_igemm_kernel for their data gradients; conv_stem_kernel for conv1)",
                        "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": round((by[0] + by[1]) / nig),
                        "algorithmic_bytes_note": "every operand of a launch once (source map, weights, destination, per-pixel "
                                                  "epilogue operands), launch-weighted mean over the same launches as `traffic`",
                        "avg_launch_us": round(1e3 * (ms[0] + ms[1]) / nig, 2),
                        "gflop_per_launch": round((fl[0] + fl[1]) / nig / 1e9, 3),
                        "sum_of_launch_ms_per_step": round(sum(ms) / args.profile_steps, 3),
                        "sum_of_launch_ms_note": "sum of per-launch event times over TWO concurrent streams (weight "
                                                 "gradients run beside the main chain): not a serial time, may exceed ms_per_step",
                        "steps_profiled": args.profile_steps, "by_kernel": per}

    if eng is None and args.model == "quadtree3d" and args.profile_steps > 0:
        # no plan behind the clip models: the Conv3d launches (qt_conv2d_igemm with 27 taps, forward + data gradient; block 1's
        # qt_conv3d_first_fwd) are timed per launch with HIP events on their stream in extra steps (video3d.py::_Ops.igemm)
        v3d = importlib.import_module(PKG + ".video3d")
        v3d.ops().timed = []
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        recs, v3d.ops().timed = v3d.ops().timed, None
        if recs:
            tot_ms = sum(a.elapsed_time(b) for a, b, _, _, _ in recs)
            tot_fl, tot_by = sum(r[2] for r in recs), sum(r[3] for r in recs)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            per = {}
            for mode, nm in ((0, "forward Conv3d launches (block 1: conv3d_first_kernel from the f32 clip; block 2: conv3d_c32_kernel, slab-resident; blocks 3-5: conv_igemm_kernel, 27 taps)"),
                             (1, "data-gradient Conv3d launches (block 2: conv3d_c32_kernel, slab-resident, two passes; blocks 3-5: conv_igemm_kernel, 27 taps)")):
                sel = [r for r in recs if r[4] == mode]
                if sel:
                    msk = sum(a.elapsed_time(b) for a, b, _, _, _ in sel)
                    per[nm] = {"launches_per_step": len(sel) // args.profile_steps, "avg_us": round(1e3 * msk / len(sel), 1),
                               "tflops": round(sum(r[2] for r in sel) / (msk * 1e-3) / 1e12, 1),
                               "algorithmic_mb_per_launch": round(sum(r[3] for r in sel) / len(sel) / 1e6, 1)}
            traffic, traffic_src = roofline_traffic(args.model, args.batch, args.dtype, ROOFLINE_FAMILIES_3D)
            roofline = {"kernel": "Conv3d launches of the step (blocks 3-5: one 27-tap implicit GEMM per convolution and direction; block 1 from the f32 clip; block 2 slab-resident)",
                        "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": round(tot_by / len(recs)), "avg_launch_us": round(1e3 * tot_ms / len(recs), 1),
                        "gflop_per_launch": round(tot_fl / len(recs) / 1e9, 3), "steps_profiled": args.profile_steps,
                        "by_kernel": per}

    # whole-step HBM traffic / matrix-pipe busy cycles from the committed PMC summaries of the same command
    # (scripts/collect_profiles.sh, scripts/collect_mfma_busy.sh) against THIS run's step time -- only from summaries
    # collected on exactly these kernel sources (otherwise null + the reason)
    hbm = mfma_pmc = None
    headline = args.batch == 256 and args.dtype == "bf16" and args.model == "quadtree" and not args.freeze_backbone
    if (headline or args.model == "quadtree3d") and not args.forward_only and not args.freeze_backbone:
        hbm = guarded("hbm summary", step_hbm, args.model, args.batch, args.dtype, elapsed / args.steps)
    if headline:
        mfma_pmc = guarded("mfma_busy summary", step_mfma_busy, args.model, args.batch, args.dtype,
                           "eval" if args.forward_only else "train", elapsed / args.steps)

    # ---- the north-star quantity on the same line: eval-mode forward of the same model, timed after the train step ----
    forward = None
    if not args.forward_only and not args.no_forward_leg:
        model.eval()

        def fwd():
            with torch.no_grad():
                return model(images, feats)
        for _ in range(3):
            fwd()
        fence()
        nf = max(10, args.steps)
        t1 = time.perf_counter()
        for _ in range(nf):
            fwd()
        fence()
        ft = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([ft], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ft = float(t.item())
        fwd_gflop = {"quadtree": FWD_GFLOP_PER_IMAGE, "attention": 3.9765, "cnn_lstm": 3.6311, "quadtree3d": 3.3814}[args.model]
        fps = B * world * nf / ft
        forward = {"metric": "images/sec eval forward (fused BN/ReLU/residual epilogues), same model and batch",
                   "value": round(fps, 1), "unit": "images/s", "ms_per_batch": round(1e3 * ft / nf, 3), "passes": nf,
                   "gflop_per_image": fwd_gflop,
                   "model_mfma_util": round(fwd_gflop * fps / world / 1e3 / MFMA_PEAK_TFLOPS[args.dtype], 4),
                   "target": "north star: >= 0.60 at bs 256 (<= 0.643 ms)" if args.model == "quadtree" else
                             "secondary line (SURVEY.md 8f): no target of its own"}
        model.train()
    # what the process group really is (a later SCALE line checks itself): backend, ranks, bytes / buckets reduced per step
    dp_info = None
    if dist is not None:
        red = getattr(model, "_grad_sync", None)
        steps_done = args.warmup + args.steps + (args.profile_steps if roofline is not None else 0)
        dp_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                   "rccl": dist.get_backend() == "nccl",
                   "gradient_bytes_per_step": int(red.bytes_reduced // max(1, steps_done)) if red is not None else None,
                   "buckets_per_step": getattr(red, "buckets_per_step", None),
                   "bucket_order": getattr(red, "bucket_log", None),
                   "exposed_tail_ms_per_step": None if exposed_tail is None else round(exposed_tail, 3),
                   "exposed_tail_note": "compute stream's wait for the communication stream at the end of backward (events on "
                                        "the compute stream around the join, rank 0, mean over the timed steps): the part of "
                                        "the gradient all-reduce that no kernel covered",
                   "average": "ncclAvg on the communication stream, joined once at the end of backward"}
    if dist is not None:
        dist.barrier()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    imgs = B * world * args.steps
    value = imgs / elapsed
    gflop_img = FWD_GFLOP_PER_IMAGE if args.forward_only else FWD_BWD_GFLOP_PER_IMAGE
    name, what = "QuadtreeCNN", "QuadtreeCNN (ResNet-18 layer3, 2x2 split, 47-feat fusion) "
    if args.model == "quadtree" and args.freeze_backbone and not args.forward_only:
        gflop_img = 3.9454  # forward + backward of the heads only
        what = "QuadtreeCNN, resnet/ variant (frozen ResNet-18 backbone, fusion mode); "
    if args.model == "attention":
        # 2*MAC of the 22 convs + 5 linears: ResNet-18 conv stack 3.6274 - fc, quadrant conv 784 px x 128x128x9,
        # sub-quadrant conv 784 px x 128x64x9, classifier 1216x1024 + 1024x12; backward = 2 x forward - conv1's dgrad
        gflop_img = 3.9765 if args.forward_only else 11.6936  # FlopCounterMode on the oracle
        name, what = "AttentionHierarchicalCNN", "AttentionHierarchicalCNN (ResNet-18 layer2 split 2x2 + 4x4, attention gate) "
    if args.model == "cnn_lstm":
        # per frame: ResNet-18 conv stack 3.6269 (forward only: the backbone is frozen) + LSTM / MLP / head ~0.004 forward,
        # x3 with their backward
        gflop_img = 3.6311 if args.forward_only else 3.6395
        name = "CnnLstm"
        what = (f"CnnLstm ({B // args.seq_len} sequences x {args.seq_len} frames per GPU, frozen per-frame ResNet-18 + pose MLP "
                "+ 2-layer LSTM(640->256) + classifier), images = frames; ")
    if args.model == "quadtree3d":
        # 2*MAC of the five Conv3d (3dcnn/models.py:108-139) on a T = 8 clip of 224x224: 2.081 + 11.098 + 5.549 + 2.774 +
        # 5.549 = 27.051 GFLOP per clip forward (+ LSTM / heads ~0.002); backward = 2 x forward - block1's data gradient
        gflop_img = 3.3814 if args.forward_only else 9.8843
        name = "Quadtree3DCNN"
        what = (f"Quadtree3DCNN (BASELINE config 4: {B // args.seq_len} clips x {args.seq_len} frames of 224x224 per GPU, five "
                "Conv3d 3x3x3 + BatchNorm3d + ReLU + MaxPool3d as 2-D MFMA launches over time-major clips, LSTM(47->188) "
                "branch, 1536->768->12 head), images = frames; ")
    out = {
        "metric": f"images/sec fwd {name} 224x224" if args.forward_only
        else f"images/sec fwd+bwd {name} 224x224 bs256",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "host_issue_ms_per_step": round(1e3 * host_issue / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic (randn images / pose vectors resident in HBM, deterministic synthetic weights)",
        "config": {"workload": what
                               + ("eval forward" if args.forward_only else
                                  ("train step fwd+bwd+Adam; " if (args.freeze_backbone or args.model in ("cnn_lstm", "quadtree3d")) else
                                   "train step fwd+bwd+Adam, all parameters trainable; ") + "Adam(lr 1e-4, wd 1e-4) by "
                                  + ("the package's FusedAdam (csrc/pack.hip)" if args.optimizer == "fused"
                                     else "torch.optim.Adam(fused=True)")),
                   "global_batch": B * world, "per_gpu_batch": B, "image": "3x224x224", "num_classes": C,
                   "parallelism": f"dp{world}"},
        "model_mfma_util": round(gflop_img * value / world / 1e3 / MFMA_PEAK_TFLOPS[args.dtype], 4),
        "forward": forward,
        "data_parallel": dp_info,
        "roofline": roofline,
        "mfma_pmc": mfma_pmc,
        "hbm": hbm,
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # the checker must never take the measurement down
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    print(json.dumps(out), flush=True)
    if dist is not None:
        os.dup2(2, 1)  # (teardown chatter, if any, also stays off stdout)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
