"""Whole-model GPU parity through the nn.Module boundary.

* against the golden vectors produced by the reference itself (tests/golden/*.npz)
* against the CPU oracle (oracle/quadtree_oracle.py) on other batch sizes.

Metric: max|x - ref| / max|ref|.  The f32-MFMA build must meet
1e-3 on logits (north-star tolerance); the bf16 throughput build is checked
against a looser, stated bound (bf16 has 8 significand bits: 4e-2).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from _util import ROOT, check_summary, pkg, rel_err

pytestmark = pytest.mark.gpu

LOGIT_TOL = {torch.float32: 1e-3, torch.bfloat16: 4e-2}
# Gradients.  Head parameters (no ReLU between them and the loss that can flip) must
# match tightly.  Backbone gradients pass through 17 ReLU masks: a pre-activation that is
# within rounding distance of zero takes the other branch under a different summation
# order (measured here: 1 element in 100,352 of layer4.0's output at B=4 in the f32 build)
# and that one flip moves every gradient below it by ~1 %.  So the f32 build is held to
# 6e-2 max-norm AND cosine >= 0.999 on the sampled elements; the bf16 build (8-bit
# significands: ~0.3 % of masks flip per layer) to cosine >= 0.85 and sums within 15 %.
HEAD_TOL = {torch.float32: 1e-4, torch.bfloat16: 2.5e-1}
BODY_TOL = {torch.float32: 6e-2, torch.bfloat16: None}
BODY_COS = {torch.float32: 0.999, torch.bfloat16: 0.85}
BUF_TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def _cos(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _oracle():
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    return o


def build(kind, dt, dropout=0.5, mode="fusion", frozen=False):
    P = pkg()
    synth = pkg("synth")
    if kind == "standard":
        m = P.StandardResNetCNN(12, dropout_rate=dropout, compute_dtype=dt)
    else:
        m = P.QuadtreeCNN(12, dropout_rate=dropout, mode=mode, freeze_backbone=frozen, compute_dtype=dt)
    m.load_state_dict(synth.synth_state_dict(m))
    return m


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_eval_logits_match_reference_golden(dt, golden_eval):
    dev = _dev()
    synth = pkg("synth")
    images = synth.synth_images(2, salt=0).to(dev)
    feats = synth.synth_pose_features(2, salt=0).to(dev)
    for name, kind, mode in [("qs_quadtree_eval", "quadtree", "fusion"), ("rn_fusion_eval", "quadtree", "fusion"),
                             ("rn_image_only_eval", "quadtree", "image_only"),
                             ("rn_numerical_only_eval", "quadtree", "numerical_only"),
                             ("rn_standard_eval", "standard", None)]:
        m = build(kind, dt, mode=mode or "fusion").to(dev).eval()
        with torch.no_grad():
            # the unused branch gets uninitialised memory, like the reference's callers do
            f = torch.empty_like(feats) if mode == "image_only" else feats
            logits = m(images, f) if kind != "standard" else m(images)
        err = rel_err(logits.cpu(), golden_eval[f"{name}/logits"])
        assert err <= LOGIT_TOL[dt], (name, err)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["qs_quadtree_train", "rn_fusion_train", "rn_standard_train"])
def test_train_step_matches_reference_golden(dt, case, golden_train):
    """train() mode (BatchNorm batch statistics), dropout p=0, CE loss, full backward."""
    dev = _dev()
    synth = pkg("synth")
    B = 4
    images = synth.synth_images(B, salt=1).to(dev)
    feats = synth.synth_pose_features(B, salt=1).to(dev)
    labels = synth.synth_labels(B, 12, salt=1).to(dev)
    if case == "rn_standard_train":
        m = build("standard", dt, dropout=0.0)
    else:
        m = build("quadtree", dt, dropout=0.0, frozen=(case == "rn_fusion_train"))
    m = m.to(dev).train()
    logits = m(images, feats)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    assert rel_err(logits.detach().cpu(), golden_train[f"{case}/logits"]) <= LOGIT_TOL[dt]
    assert abs(loss.item() - float(golden_train[f"{case}/loss"])) <= LOGIT_TOL[dt] * max(1.0, abs(float(golden_train[f"{case}/loss"]))) * 10
    names = list(golden_train[f"{case}/grad_names"])
    params = dict(m.named_parameters())
    got_names = [n for n, p in params.items() if p.grad is not None]
    assert sorted(got_names) == sorted(names)
    from _util import summary
    for n in names:
        g = params[n].grad.detach().cpu()
        pre = f"{case}/grad/{n}"
        assert tuple(g.shape) == tuple(int(x) for x in golden_train[f"{pre}/shape"])
        smp, gold = summary(g)["sample"], golden_train[f"{pre}/sample"]
        err = float(np.abs(smp - gold).max()) / max(float(np.abs(gold).max()), 1e-30)
        if n.startswith("base_cnn."):
            assert _cos(smp, gold) >= BODY_COS[dt], (n, _cos(smp, gold))
            if BODY_TOL[dt] is not None:
                assert err <= BODY_TOL[dt], (n, err)
            else:
                gs = float(golden_train[f"{pre}/abssum"])
                assert abs(summary(g)["abssum"] - gs) <= 0.15 * gs, n
        else:
            assert err <= HEAD_TOL[dt], (n, err)
    # running statistics were updated with torch's momentum / unbiased-variance rule
    bufs = dict(m.named_buffers())
    for k in golden_train.files:
        if k.startswith(f"{case}/buf/") and k.endswith("/shape"):
            n = k[len(f"{case}/buf/"):-len("/shape")]
            check_summary(bufs[n].cpu(), golden_train, f"{case}/buf/{n}", BUF_TOL[dt])
    assert int(bufs["base_cnn.bn1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("dt", [torch.float32])
def test_eval_matches_oracle_other_batches(dt):
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    m = build("quadtree", dt).to(dev).eval()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    for B in (1, 5):
        images = synth.synth_images(B, salt=7 + B)
        feats = synth.synth_pose_features(B, salt=7 + B)
        with torch.no_grad():
            ref = o.quadtree_forward(sd, images, feats)
            got = m(images.to(dev), feats.to(dev)).cpu()
        assert rel_err(got, ref) <= LOGIT_TOL[dt]


def test_cpu_tensors_are_rejected_loudly():
    _dev()
    m = build("quadtree", torch.float32)
    with pytest.raises(pkg().QtError):
        m(torch.zeros(1, 3, 224, 224), torch.zeros(1, 47))


def test_gradcam_hooks_on_layer4_match_oracle():
    """The reference's Grad-CAM recipe (resnet/grad_cam_analysis.py:243-316): eval(), forward hook +
    full backward hook on model.base_cnn.layer4, one-hot backward.  Frozen-backbone variant."""
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    m = build("quadtree", torch.float32, frozen=True).to(dev).eval()
    h1 = m.base_cnn.layer4.register_forward_hook(m.save_activation_hook)
    h2 = m.base_cnn.layer4.register_full_backward_hook(m.save_gradient_hook)
    B = 2
    x, f = synth.synth_images(B, salt=11), synth.synth_pose_features(B, salt=11)
    xi = x.to(dev).requires_grad_(True)
    logits = m(xi, f.to(dev))
    one_hot = torch.zeros_like(logits)
    one_hot[:, 3] = 1.0
    logits.backward(gradient=one_hot, retain_graph=True)
    torch.cuda.synchronize()
    h1.remove()
    h2.remove()
    assert m.activations.shape == (B, 512, 7, 7) and m.gradients.shape == (B, 512, 7, 7)
    # oracle: same quantities with torch autograd on CPU
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    taps = {}
    ref = o.quadtree_forward(sd, x.requires_grad_(True), f, taps=taps)
    taps["layer4"].retain_grad()
    ref.backward(gradient=one_hot.cpu())
    assert rel_err(m.activations.cpu(), taps["layer4"].detach()) <= 1e-4
    assert rel_err(m.gradients.cpu(), taps["layer4"].grad) <= 1e-4
    # the Grad-CAM map itself (pooled-gradient weighting, grad_cam_analysis.py:306-316)
    cam = lambda a, g: torch.relu((g.mean((2, 3), keepdim=True) * a).sum(1))
    assert rel_err(cam(m.activations.cpu(), m.gradients.cpu()), cam(taps["layer4"].detach(), taps["layer4"].grad)) <= 1e-3


def test_three_optimizer_steps_follow_the_oracle_trajectory():
    """The reference's hot loop (Quadtree_train.py:62-66) for three steps, ragged batch of 6,
    f32 build, dropout 0: losses must track the CPU oracle step by step (checks weight
    re-packing after optimizer.step, running-stat updates, gradient hand-over to the optimizer).
    SGD+momentum instead of the reference's Adam: Adam divides by sqrt(v), so parameters whose
    gradient is at rounding level move by ~lr in a rounding-determined direction and the two
    trajectories separate chaotically (measured 3e-4 after one step, 3e-3 after two)."""
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    B = 6
    m = build("quadtree", torch.float32, dropout=0.0)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    opt = torch.optim.SGD(m.parameters(), lr=2e-4, momentum=0.9, weight_decay=1e-4)
    keys = o.trainable_keys(sd0, False)
    sd = o.unique_params(sd0, keys)
    opt_ref = torch.optim.SGD([sd[k] for k in keys], lr=2e-4, momentum=0.9, weight_decay=1e-4)
    got, ref = [], []
    for step in range(3):
        x = synth.synth_images(B, salt=20 + step)
        f = synth.synth_pose_features(B, salt=20 + step)
        y = synth.synth_labels(B, 12, salt=20 + step)
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x.to(dev), f.to(dev)), y.to(dev))
        loss.backward()
        opt.step()
        got.append(loss.item())
        opt_ref.zero_grad()
        lr = torch.nn.functional.cross_entropy(o.quadtree_forward(sd, x, f, train=True, dropout_p=0.0), y)
        lr.backward()
        opt_ref.step()
        ref.append(lr.item())
    for a, b in zip(got, ref):
        assert abs(a - b) <= 3e-3 * max(1.0, abs(b)), (got, ref)
    # eval after training uses the updated running statistics
    m.eval()
    x, f = synth.synth_images(2, salt=30), synth.synth_pose_features(2, salt=30)
    with torch.no_grad():
        e1 = m(x.to(dev), f.to(dev)).cpu()
        e2 = o.quadtree_forward(sd, x, f)
    assert rel_err(e1, e2) <= 5e-3


@pytest.mark.parametrize("opt_kw", [dict(fused=True), dict(foreach=True), dict(foreach=False)])
def test_updated_weights_are_used_after_fused_optimizer_step(opt_kw):
    """torch's fused Adam updates parameters without bumping Tensor._version, so the plan must not
    key its packed operand copies on the version counters alone: after optimizer.step() the next
    forward has to see the new weights (Quadtree_train.py:62-66 relies on that every step)."""
    dev = _dev()
    synth = pkg("synth")
    B = 4
    m = build("quadtree", torch.bfloat16, dropout=0.0).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, **opt_kw)
    x, f = synth.synth_images(B, salt=40).to(dev), synth.synth_pose_features(B, salt=40).to(dev)
    y = synth.synth_labels(B, 12, salt=40).to(dev)
    m.eval()
    with torch.no_grad():
        before = m(x, f).clone()
    m.train()
    opt.zero_grad()
    torch.nn.functional.cross_entropy(m(x, f), y).backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        after = m(x, f).clone()
    # a fresh model loaded with the updated state_dict is the ground truth for "the new weights"
    fresh = build("quadtree", torch.bfloat16, dropout=0.0)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    fresh = fresh.to(dev).eval()
    with torch.no_grad():
        want = fresh(x, f)
    assert not torch.equal(before, after), "the optimizer step did not reach the forward pass"
    assert torch.equal(after, want)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batch_growth_and_dropout_statistics(dt):
    """The engine is rebuilt when the batch outgrows the plan; dropout keeps about half of the
    hidden units in train() and is the identity in eval()."""
    dev = _dev()
    m = build("quadtree", dt, dropout=0.5).to(dev)
    m.eval()
    with torch.no_grad():
        a = m(torch.randn(3, 3, 224, 224, device=dev), torch.randn(3, 47, device=dev))
        b = m(torch.randn(9, 3, 224, 224, device=dev), torch.randn(9, 47, device=dev))
    assert a.shape == (3, 12) and b.shape == (9, 12) and m._engine.max_batch >= 9
    m.train()
    x, f = torch.randn(8, 3, 224, 224, device=dev), torch.randn(8, 47, device=dev)
    torch.manual_seed(1)
    l1 = m(x, f)
    hid = m._engine.buffer("hidden", (m._engine.max_batch, 2688))[:8].float()
    frac_zero = float((hid == 0).float().mean())
    assert 0.55 <= frac_zero <= 0.95  # ReLU zeros plus ~half of the rest dropped
    torch.manual_seed(2)
    l2 = m(x, f)
    assert not torch.allclose(l1, l2)  # different dropout masks
    torch.manual_seed(1)
    l3 = m(x, f)
    # same seed -> same mask (running stats moved a little, so compare loosely)
    assert rel_err(l3.detach().cpu(), l1.detach().cpu()) < rel_err(l2.detach().cpu(), l1.detach().cpu())


def test_fused_adam_matches_torch_adam_and_repacks():
    """FusedAdam (csrc/pack.hip: optimizer step inside the one-launch weight packing + multi-tensor
    kernel) against torch.optim.Adam with the reference's hyper-parameters
    (Quadtree_train.py:45: lr 1e-4, weight_decay 1e-4), three steps on identical models."""
    dev = _dev()
    P = pkg()
    synth = pkg("synth")
    B = 4
    a = build("quadtree", torch.bfloat16, dropout=0.0).to(dev).train()
    b = build("quadtree", torch.bfloat16, dropout=0.0).to(dev).train()
    oa = torch.optim.Adam(a.parameters(), lr=1e-4, weight_decay=1e-4, foreach=False)
    ob = P.FusedAdam(b.parameters(), lr=1e-4, weight_decay=1e-4, model=b)
    for step in range(3):
        x, f = synth.synth_images(B, salt=50 + step).to(dev), synth.synth_pose_features(B, salt=50 + step).to(dev)
        y = synth.synth_labels(B, 12, salt=50 + step).to(dev)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            torch.nn.functional.cross_entropy(m(x, f), y).backward()
            o.step()
        if step == 0:
            # identical weights going in; the gradients agree to summation order (the generic weight-gradient
            # kernels accumulate with float atomics), so the states agree to ~1e-5, not to the last bit
            # (the update rule itself is pinned to 2e-6 by test_adam_multi_kernel_matches_torch_single_tensor_adam)
            pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
            for k in pa:
                if pa[k].grad is None:
                    continue
                assert rel_err(pb[k].detach().cpu(), pa[k].detach().cpu()) <= 1e-4, k
                sa, sb = oa.state[pa[k]], ob.state[pb[k]]
                assert rel_err(sb["exp_avg"].cpu(), sa["exp_avg"].cpu()) <= 1e-4, k
                assert rel_err(sb["exp_avg_sq"].cpu(), sa["exp_avg_sq"].cpu()) <= 1e-4, k
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in ("classifier.0.weight", "base_cnn.layer1.0.conv1.weight", "base_cnn.conv1.weight", "classifier.3.bias",
              "base_cnn.layer4.1.bn2.weight", "numerical_mlp.0.weight"):
        # Adam moves a parameter by ~lr per step in the direction sign(m): where the gradient is at rounding
        # level that sign follows the summation order, so two correct runs may part by up to 2*lr per step
        assert float((pb[k].detach() - pa[k].detach()).abs().max()) <= 2 * 1e-4 * 3 + 1e-6, k
    # the packed operand copies were refreshed by the step itself
    b.eval()
    x, f = synth.synth_images(2, salt=60).to(dev), synth.synth_pose_features(2, salt=60).to(dev)
    with torch.no_grad():
        got = b(x, f).clone()
    fresh = build("quadtree", torch.bfloat16, dropout=0.0)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in b.state_dict().items()})
    fresh = fresh.to(dev).eval()
    with torch.no_grad():
        want = fresh(x, f)
    assert torch.equal(got, want)


def test_adam_multi_kernel_matches_torch_single_tensor_adam():
    dev = _dev()
    L = pkg("_lib")
    eng = pkg("engine")
    lib = L.lib()
    g = torch.Generator().manual_seed(61)
    shapes = [(5,), (4096,), (4097, 3), (64, 3, 7, 7), (1,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=3e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.02, foreach=False)
    mine = [p.clone().to(dev) for p in ps]
    ms = [torch.zeros_like(p) for p in mine]
    vs = [torch.zeros_like(p) for p in mine]
    lib.qt_adam_multi.argtypes = [ctypes.POINTER(eng.AdamItem), ctypes.c_int, ctypes.POINTER(eng.AdamDesc), ctypes.c_void_p]
    for step in range(1, 4):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
        gd = [gr.to(dev) for gr in grads]
        items = (eng.AdamItem * len(mine))(*[eng.AdamItem(p.data_ptr(), q.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                                            for p, q, m, v in zip(mine, gd, ms, vs)])
        desc = eng.AdamDesc(3e-3, 0.8, 0.95, 1e-6, 0.02, 1.0, step)
        L.check(lib.qt_adam_multi(items, len(mine), ctypes.byref(desc), L.stream_ptr()), "qt_adam_multi")
        torch.cuda.synchronize()
    for r, p, m, v in zip(ref, mine, ms, vs):
        assert rel_err(p.cpu(), r.detach()) <= 2e-6
        assert rel_err(m.cpu(), opt.state[r]["exp_avg"]) <= 2e-6
        assert rel_err(v.cpu(), opt.state[r]["exp_avg_sq"]) <= 2e-6


@pytest.mark.parametrize("mode", ["image_only", "numerical_only"])
def test_frozen_backbone_modes_train_step_matches_oracle(mode):
    """resnet/ variant with TRAINING_MODE image_only / numerical_only (resnet/models.py:115-129,
    resnet/train_cnn_model.py:15): frozen backbone, train-mode BatchNorm, dropout 0, f32 build.
    Loss and every trainable parameter's gradient against the CPU oracle; the branch the mode ignores
    gets uninitialised input, as the reference's callers pass it."""
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    B = 3
    m = build("quadtree", torch.float32, dropout=0.0, mode=mode, frozen=True)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    x, f = synth.synth_images(B, salt=70), synth.synth_pose_features(B, salt=70)
    y = synth.synth_labels(B, 12, salt=70)
    xin = x.to(dev) if mode != "numerical_only" else torch.empty(B, 3, 224, 224, device=dev)
    fin = f.to(dev) if mode != "image_only" else torch.empty(B, 47, device=dev)
    loss = torch.nn.functional.cross_entropy(m(xin, fin), y.to(dev))
    loss.backward()
    keys = [k for k in o.trainable_keys(sd0, True)
            if not (mode == "image_only" and k.startswith("numerical_mlp."))
            and not (mode == "numerical_only" and k.startswith("quadrant_processor."))]
    sd = o.unique_params(sd0, keys)
    ref_loss = torch.nn.functional.cross_entropy(o.quadtree_forward(sd, x, f, mode=mode, train=True, dropout_p=0.0), y)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-3 * max(1.0, abs(ref_loss.item()))
    got = dict(m.named_parameters())
    checked = 0
    for k in keys:
        if sd[k].grad is None:
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k
            continue
        assert got[k].grad is not None, k
        assert rel_err(got[k].grad.cpu(), sd[k].grad) <= 1e-3, k
        checked += 1
    assert checked >= 4
    for k, p in got.items():   # the frozen backbone hands out no gradients
        if k.startswith("base_cnn.") or k.startswith("features_extractor.") or k.startswith("global_processor."):
            assert p.grad is None, k


def test_frozen_weights_are_not_repacked_by_adam_step_but_follow_external_changes():
    """qt_plan_adam_step leaves the packed copies of FROZEN weights alone (they did not change); when the user replaces
    them (load_state_dict bumps the version counters) the next forward must run on the new values."""
    dev = _dev()
    P, synth = pkg(), pkg("synth")
    B = 4
    m = build("quadtree", torch.bfloat16, dropout=0.0, frozen=True).to(dev).train()
    opt = P.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-4, model=m)
    x, f = synth.synth_images(B, salt=70).to(dev), synth.synth_pose_features(B, salt=70).to(dev)
    y = synth.synth_labels(B, 12, salt=70).to(dev)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x, f), y).backward()
        opt.step()
    assert m.base_cnn.conv1.weight.grad is None
    # after two steps the model equals a fresh one loaded with its state: frozen copies current, trained ones re-packed
    m.eval()
    with torch.no_grad():
        got = m(x, f).clone()
    fresh = build("quadtree", torch.bfloat16, dropout=0.0, frozen=True)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    fresh = fresh.to(dev).eval()
    with torch.no_grad():
        want = fresh(x, f)
    assert torch.equal(got, want)
    # replace the frozen backbone's weights behind the optimizer's back
    new = synth.synth_state_dict(m, salt=1)
    m.load_state_dict(new)
    other = build("quadtree", torch.bfloat16, dropout=0.0, frozen=True)
    other.load_state_dict(new)
    other = other.to(dev).eval()
    with torch.no_grad():
        assert torch.equal(m(x, f), other(x, f))
    m.train()
    opt.zero_grad()
    torch.nn.functional.cross_entropy(m(x, f), y).backward()
    opt.step()
    m.eval()
    again = build("quadtree", torch.bfloat16, dropout=0.0, frozen=True)
    again.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    again = again.to(dev).eval()
    with torch.no_grad():
        assert torch.equal(m(x, f), again(x, f))


@pytest.mark.parametrize("kind", ["quadtree", "attention"])
def test_eval_mode_backward_through_trainable_backbone(kind):
    """model.eval() + logits.backward() with every parameter trainable: the Grad-CAM recipe on the reference's all-trainable
    variant (Quadtree_from scratch/grad_cam.py:72-83).  The forward then runs eval arithmetic (running statistics, no dropout)
    but keeps what backward needs (qt_plan_forward training = 2); BatchNorm backward has no batch-mean terms.  Logits must
    equal the fused eval forward; hooks and every parameter gradient are checked against the oracle in eval mode."""
    dev = _dev()
    o = _oracle()
    P, synth = pkg(), pkg("synth")
    B = 3
    x, f = synth.synth_images(B, salt=15), synth.synth_pose_features(B, salt=15)
    if kind == "quadtree":
        m = build("quadtree", torch.float32, dropout=0.5)
    else:
        m = P.AttentionHierarchicalCNN(12, dropout_rate=0.5, compute_dtype=torch.float32)
        m.load_state_dict(synth.synth_state_dict(m))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).eval()
    with torch.no_grad():
        fused = m(x.to(dev), f.to(dev)).clone()
    if kind == "quadtree":
        h1 = m.base_cnn.layer4.register_forward_hook(m.save_activation_hook)
        h2 = m.base_cnn.layer4.register_full_backward_hook(m.save_gradient_hook)
    logits = m(x.to(dev), f.to(dev))
    assert rel_err(logits.detach().cpu(), fused.cpu()) <= 1e-5
    one_hot = torch.zeros_like(logits)
    one_hot[:, 5] = 1.0
    m.zero_grad()
    logits.backward(gradient=one_hot)
    torch.cuda.synchronize()
    # running statistics are untouched by an eval forward
    bn_key = "base_cnn.bn1.running_mean" if kind == "quadtree" else "features_extractor.1.running_mean"
    assert torch.equal(m.state_dict()[bn_key].cpu(), sd0[bn_key])
    if kind == "quadtree":
        keys = o.trainable_keys(sd0, False)
        sd = o.unique_params(sd0, keys)
        taps = {}
        ref = o.quadtree_forward(sd, x, f, train=False, taps=taps)
        taps["layer4"].retain_grad()
        ref.backward(gradient=one_hot.cpu())
        want = {k: sd[k].grad for k in keys}
        h1.remove()
        h2.remove()
        assert rel_err(m.activations.cpu(), taps["layer4"].detach()) <= 1e-4
        assert rel_err(m.gradients.cpu(), taps["layer4"].grad) <= 1e-4
    else:
        names = [n for n, _ in m.named_parameters()]
        leaves = {n: sd0[n].clone().requires_grad_(True) for n in names}
        sdb = o.attention_sd_to_base({k: leaves.get(k, v.clone()) for k, v in sd0.items()})
        ref = o.attention_forward(sdb, x, f, train=False)
        ref.backward(gradient=one_hot.cpu())
        want = {n: leaves[n].grad for n in names}
    assert rel_err(logits.detach().cpu(), ref.detach()) <= 1e-3
    params = dict(m.named_parameters())
    from _util import summary
    for n, gref in want.items():
        assert params[n].grad is not None, n
        got, ref_s = summary(params[n].grad.detach().cpu(), 2048)["sample"], summary(gref, 2048)["sample"]
        if n == "attention_gate.2.bias":
            assert float(np.abs(got).max()) <= 1e-5
            continue
        err = float(np.abs(got - ref_s).max()) / max(float(np.abs(ref_s).max()), 1e-30)
        body = n.startswith(("base_cnn.", "features_extractor.", "global_processor.", "quadrant_processor.",
                             "sub_quadrant_processor."))
        if body:   # ReLU decisions at rounding distance from zero (see HEAD_TOL / BODY_TOL above)
            assert _cos(got, ref_s) >= 0.999 and err <= 6e-2, (n, err, _cos(got, ref_s))
        else:
            assert err <= 1e-4, (n, err)


@pytest.mark.parametrize("env", [{"QTCNN_S2_DGRAD_MERGED": "0"}, {"QTCNN_WP_VARIANT": "0", "QTCNN_PT_CONV": "0"}])
def test_alternate_kernel_selections_keep_train_step_parity(env):
    """The kernel-selection switches are read once per process: the round-1 forms they keep reachable (four parity-class
    gathers for the stride-2 data gradients; ring weight-gradient kernel and generic 3x3 convs) are exercised by running
    the reference-golden train-step test of the f32 parity build in a child process with the switch set."""
    import subprocess
    child_env = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_train_step_matches_reference_golden and qs_quadtree_train"],
                       env=child_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "2 passed" in r.stdout, r.stdout[-500:]
