"""GPU parity at BASELINE.json's full size (config 2: QuadtreeCNN, batch 256, 224x224).

* the f32-MFMA build against the CPU oracle on the SAME 256-image batch: one train-mode step (BatchNorm statistics over all
  256 images, dropout 0): logits <= 1e-3 (north-star tolerance), loss, pose-MLP / classifier gradients <= 1e-3 with
  cosine >= 0.999999, backbone and quadrant-conv gradients by the ReLU-flip-aware rule of test_model_gpu.py.  The oracle needs ~10 s of the box's host cores for this batch.
* size-independent properties of the bf16 throughput build at batch 256:
    - eval forward is per-image: the batch-256 logits equal the logits of the same images run 32 at a time;
    - backward is linear in d(loss)/d(logits): scaling the loss by 2 scales every gradient by 2 (a power of two: only the
      order of the float atomics in the generic weight-gradient kernel may differ);
    - rows of softmax-cross-entropy gradients sum to zero, so classifier.3.bias.grad sums to ~0;
    - gradients exist for exactly the 70 used parameters (not for base_cnn.fc).
"""
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import ROOT, pkg, rel_err, summary

pytestmark = pytest.mark.gpu
B = 256


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _build(dt):
    P, synth = pkg(), pkg("synth")
    m = P.QuadtreeCNN(12, dropout_rate=0.0, compute_dtype=dt, max_batch=B)
    m.load_state_dict(synth.synth_state_dict(m))
    return m


def _cos(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def test_f32_train_step_at_batch_256_matches_oracle():
    dev = _dev()
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    synth = pkg("synth")
    torch.set_num_threads(16)
    x, f, y = synth.synth_images(B, salt=77), synth.synth_pose_features(B, salt=77), synth.synth_labels(B, 12, salt=77)
    m = _build(torch.float32)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    logits = m(x.to(dev), f.to(dev))
    loss = F.cross_entropy(logits, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    keys = o.trainable_keys(sd0, False)
    sd = o.unique_params(sd0, keys)
    ref = o.quadtree_forward(sd, x, f, train=True, dropout_p=0.0)
    ref_loss = F.cross_entropy(ref, y)
    ref_loss.backward()
    assert rel_err(logits.detach().cpu(), ref.detach()) <= 1e-3
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    params = dict(m.named_parameters())
    assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(keys)
    # Pure head parameters (pose MLP, classifier) see the image branch only through the fused features; the quadrant conv's
    # weight gradient is a product with layer3's output map itself, where the handful of ReLU decisions that differ between
    # two summation orders (see test_model_gpu.py) enter directly: it is held to the backbone rule.
    report, bad = [], []
    for n in keys:
        got, want = summary(params[n].grad.detach().cpu(), 4096)["sample"], summary(sd[n].grad, 4096)["sample"]
        err = float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)
        cos = _cos(got, want)
        report.append((n, err, cos))
        if n.startswith(("base_cnn.", "quadrant_processor.")):
            ok = cos >= 0.999 and err <= 6e-2
        else:
            ok = err <= 1e-3 and cos >= 0.999999
        if not ok:
            bad.append((n, err, cos))
    heads = [r for r in report if not r[0].startswith("base_cnn.")]
    print("B=256 f32 head gradients:", [(n, f"{e:.1e}") for n, e, _ in heads])
    print("B=256 f32 backbone: max err %.2e, min cosine %.6f" % (max(r[1] for r in report if r[0].startswith("base_cnn.")),
                                                                 min(r[2] for r in report if r[0].startswith("base_cnn."))))
    assert not bad, bad
    # the updated BatchNorm running statistics (momentum 0.1, unbiased variance over 256 x H x W values)
    bufs = dict(m.named_buffers())
    for n in ("base_cnn.bn1.running_var", "base_cnn.layer3.1.bn2.running_mean", "base_cnn.layer4.0.downsample.1.running_var"):
        assert rel_err(bufs[n].cpu(), sd[n]) <= 1e-4, n
    # The bf16 throughput build on the SAME batch against the SAME oracle logits: its error at BASELINE config 2's size
    # is measured and bounded here (3.0e-3 at B = 4 on the fixtures; the bound is the one the fixtures use), and its
    # loss must agree with the oracle's.  Eval-mode logits (running statistics) against the oracle's eval forward too.
    del m
    torch.cuda.empty_cache()
    mb = _build(torch.bfloat16).to(dev).train()
    lb = mb(x.to(dev), f.to(dev))
    loss_b = F.cross_entropy(lb, y.to(dev))
    err_b = rel_err(lb.detach().float().cpu(), ref.detach())
    mb2 = _build(torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        le = mb2(x.to(dev), f.to(dev)).float().cpu()
        ref_e = o.quadtree_forward({k: v.detach() for k, v in sd0.items()}, x, f)
    err_e = rel_err(le, ref_e)
    print("B=256 bf16 logits vs oracle: train-mode %.2e, eval %.2e (max|d| / max|ref|); loss %.5f vs %.5f"
          % (err_b, err_e, loss_b.item(), ref_loss.item()))
    assert err_b <= 4e-2 and err_e <= 4e-2, (err_b, err_e)
    assert abs(loss_b.item() - ref_loss.item()) <= 2e-2 * max(1.0, abs(ref_loss.item()))
    assert (le.argmax(1) == ref_e.argmax(1)).float().mean().item() >= 0.97


def test_bf16_properties_at_batch_256():
    dev = _dev()
    synth = pkg("synth")
    x, f = synth.synth_images(B, salt=78).to(dev), synth.synth_pose_features(B, salt=78).to(dev)
    y = synth.synth_labels(B, 12, salt=78).to(dev)
    m = _build(torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        full = m(x, f).clone()
        parts = torch.cat([m(x[i:i + 32], f[i:i + 32]).clone() for i in range(0, B, 32)])
    assert rel_err(parts.cpu(), full.cpu()) <= 1e-6   # per-image arithmetic does not depend on the batch around it

    m.train()
    grads = []
    for scale in (1.0, 2.0):
        for p in m.parameters():
            p.grad = None
        # same batch statistics both times: put the running statistics back (they do not enter train-mode outputs)
        (F.cross_entropy(m(x, f), y) * scale).backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert len(grads[0]) == 70 and not any(n.startswith("base_cnn.fc.") for n in grads[0])
    for n, g1 in grads[0].items():
        g2 = grads[1][n]
        assert float((g2 - 2 * g1).abs().max()) <= 2e-5 * max(float(g2.abs().max()), 1e-30), n
    b3 = grads[0]["classifier.3.bias"]
    assert abs(float(b3.sum())) <= 1e-5 * max(float(b3.abs().max()), 1e-30) + 1e-7
