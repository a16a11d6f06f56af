"""GPU parity of the CnnLstm path (reference cnn+lstm/models.py:14-89): the LSTM
recurrence kernels of csrc/lstm.hip through the C ABI against the oracle's gate-by-gate restatement, and the
whole model against the vectors the reference produced with torch's nn.LSTM (tests/golden/cnn_lstm_b2t3.npz)
and against the CPU oracle at another (B, T).

Tolerances: f32 build 1e-3 on logits (measured ~1e-6), head gradients 1e-4 (no ReLU mask between them and
the loss can flip: the frozen backbone takes no gradient); bf16 build 4e-2 on logits as for the other models.
"""
import ctypes
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import ROOT, check_summary, pkg, rel_err, summary

pytestmark = pytest.mark.gpu
LOGIT_TOL = {torch.float32: 1e-3, torch.bfloat16: 4e-2}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _oracle():
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    return o


@pytest.mark.parametrize("H,I", [(256, 640), (64, 47)])
def test_lstm_layer_forward_backward(H, I):
    """one nn.LSTM layer: forward states and the gradient w.r.t. the pre-activation gates (-> dx, dW_ih, dW_hh, db)"""
    dev = _dev()
    o = _oracle()
    L = pkg("_lib")
    lib = L.lib()
    B, T = 5, 6
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, T, I, generator=g).requires_grad_(True)
    sd = {"lstm.weight_ih_l0": (torch.randn(4 * H, I, generator=g) / I ** 0.5).requires_grad_(True),
          "lstm.weight_hh_l0": (torch.randn(4 * H, H, generator=g) / H ** 0.5).requires_grad_(True),
          "lstm.bias_ih_l0": (torch.randn(4 * H, generator=g) * 0.1).requires_grad_(True),
          "lstm.bias_hh_l0": (torch.randn(4 * H, generator=g) * 0.1).requires_grad_(True)}
    ref = o._lstm_layer(sd, 0, x)
    dh = torch.randn(B, T, H, generator=g)
    dlast = torch.randn(B, H, generator=g)
    (ref * dh).sum().add((ref[:, -1] * dlast).sum()).backward()
    f32 = dict(dtype=torch.float32, device=dev)
    w_ih, w_hh, b_ih, b_hh = (sd[k].detach().to(dev) for k in ("lstm.weight_ih_l0", "lstm.weight_hh_l0",
                                                                "lstm.bias_ih_l0", "lstm.bias_hh_l0"))
    xd = x.detach().to(dev)
    xproj = (xd.view(B * T, I) @ w_ih.t() + b_ih).contiguous()
    whh_t = torch.empty(H, 4 * H, **f32)
    L.check(lib.qt_transpose_f32(L.ptr(w_hh), L.ptr(whh_t), 4 * H, H, L.stream_ptr()), "qt_transpose_f32")
    assert torch.equal(whh_t, w_hh.t())
    gates, cell = torch.empty(B, T, 4 * H, **f32), torch.empty(B, T, H, **f32)
    hprev, hout = torch.empty(B, T, H, **f32), torch.empty(B, T, H, **f32)
    L.check(lib.qt_lstm_forward(L.ptr(xproj), L.ptr(whh_t), L.ptr(b_hh), L.ptr(gates), L.ptr(cell), L.ptr(hprev),
                                L.ptr(hout), B, T, H, L.stream_ptr()), "qt_lstm_forward")
    assert rel_err(hout.cpu(), ref.detach()) <= 2e-5
    assert torch.equal(hprev[:, 1:], hout[:, :-1]) and float(hprev[:, 0].abs().max()) == 0.0
    dgates = torch.empty(B, T, 4 * H, **f32)
    dhd, dld = dh.to(dev), dlast.to(dev)
    L.check(lib.qt_lstm_backward(L.ptr(dhd), L.ptr(dld), L.ptr(gates), L.ptr(cell), L.ptr(w_hh), L.ptr(dgates), B, T, H,
                                 L.stream_ptr()), "qt_lstm_backward")
    dG = dgates.view(B * T, 4 * H)
    assert rel_err((dG @ w_ih).cpu().view(B, T, I), x.grad) <= 5e-5
    assert rel_err((dG.t() @ xd.view(B * T, I)).cpu(), sd["lstm.weight_ih_l0"].grad) <= 5e-5
    assert rel_err((dG.t() @ hprev.view(B * T, H)).cpu(), sd["lstm.weight_hh_l0"].grad) <= 5e-5
    assert rel_err(dG.sum(0).cpu(), sd["lstm.bias_ih_l0"].grad) <= 5e-5
    assert rel_err(dG.sum(0).cpu(), sd["lstm.bias_hh_l0"].grad) <= 5e-5


def _build(dt, T, dropout=0.0):
    P, synth = pkg(), pkg("synth")
    m = P.CnnLstm(12, sequence_length=T, dropout_rate=dropout, compute_dtype=dt)
    m.load_state_dict(synth.synth_state_dict(m))
    return m


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cnn_lstm_matches_reference_golden(dt, golden_cnn_lstm):
    dev = _dev()
    synth = pkg("synth")
    g = golden_cnn_lstm
    B, T = 2, 3
    x = synth.synth_images(B * T, salt=7).view(B, T, 3, 224, 224).to(dev)
    f = synth.synth_pose_features(B * T, salt=7).view(B, T, 47).to(dev)
    y = synth.synth_labels(B, 12, salt=7).to(dev)
    m = _build(dt, T).to(dev).eval()
    with torch.no_grad():
        logits = m(x, f)
    assert tuple(logits.shape) == (B, 12)
    err = rel_err(logits.cpu(), g["eval/logits"])
    assert err <= LOGIT_TOL[dt], err
    fused = m._engine.buffer("fused", (m._engine.max_batch, 640))[:B * T].float().cpu().view(B, T, 640)
    check_summary(fused, g, "eval/tap/fused", 1e-4 if dt == torch.float32 else 4e-2)

    m.train()
    logits = m(x, f)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    assert rel_err(logits.detach().cpu(), g["train/logits"]) <= LOGIT_TOL[dt]
    gl = float(g["train/loss"])
    assert abs(loss.item() - gl) <= LOGIT_TOL[dt] * max(1.0, abs(gl)) * 10
    names = list(g["train/grad_names"])
    params = dict(m.named_parameters())
    assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(names)
    for n in names:
        gr = params[n].grad.detach().cpu()
        pre = f"train/grad/{n}"
        assert tuple(gr.shape) == tuple(int(v) for v in g[f"{pre}/shape"])
        smp, gold = summary(gr)["sample"], g[f"{pre}/sample"]
        err = float(np.abs(smp - gold).max()) / max(float(np.abs(gold).max()), 1e-30)
        assert err <= (1e-4 if dt == torch.float32 else 2.5e-1), (n, err)
    bufs = dict(m.named_buffers())
    for k in g.files:
        if k.startswith("train/buf/") and k.endswith("/shape"):
            n = k[len("train/buf/"):-len("/shape")]
            check_summary(bufs[n].cpu(), g, f"train/buf/{n}", 1e-4 if dt == torch.float32 else 2e-2)


def test_cnn_lstm_matches_oracle_other_shape_and_trains():
    """f32 build, 3 sequences of 4 frames (the reference's SEQ_LEN): eval logits, then two Adam steps lower the loss"""
    dev = _dev()
    o = _oracle()
    P, synth = pkg(), pkg("synth")
    B, T = 3, 4
    x = synth.synth_images(B * T, salt=13).view(B, T, 3, 224, 224)
    f = synth.synth_pose_features(B * T, salt=13).view(B, T, 47)
    y = synth.synth_labels(B, 12, salt=13)
    m = _build(torch.float32, T).to(dev).eval()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        got = m(x.to(dev), f.to(dev)).cpu()
        ref = o.cnn_lstm_forward(o.cnn_lstm_sd_to_base(sd0), x, f)
    assert rel_err(got, ref) <= 1e-3
    m.train()
    opt = P.FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3, model=m)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = F.cross_entropy(m(x.to(dev), f.to(dev)), y.to(dev))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses
    # train-mode dropout (p = 0.5) runs and keeps the output finite
    d = _build(torch.bfloat16, T, dropout=0.5).to(dev).train()
    out = d(x.to(dev), f.to(dev))
    F.cross_entropy(out, y.to(dev)).backward()
    assert torch.isfinite(out).all() and torch.isfinite(d.lstm.weight_ih_l0.grad).all()


def test_cnn_lstm_config5_size_t16_x6_matches_oracle():
    """BASELINE config 5 at its own size: 6 viewpoint sequences x T = 16 frames (cnn+lstm/models.py:58-89 on
    [6,16,3,224,224] + [6,16,47]).  f32 build: eval logits and a dropout-free train step (loss + all 16 trainable
    gradients) against the CPU oracle; the 16-step recurrence is 4x longer than the fixtures' T=3/4."""
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    B, T = 6, 16
    x = synth.synth_images(B * T, salt=160).view(B, T, 3, 224, 224)
    f = synth.synth_pose_features(B * T, salt=160).view(B, T, 47)
    y = synth.synth_labels(B, 12, salt=160)
    m = _build(torch.float32, T).to(dev).eval()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    base = o.cnn_lstm_sd_to_base(sd0)
    with torch.no_grad():
        got = m(x.to(dev), f.to(dev)).cpu()
        ref = o.cnn_lstm_forward(base, x, f)
    assert got.shape == (B, 12)
    assert rel_err(got, ref) <= 1e-3
    # train step (train-mode BatchNorm over the 96 frames, dropout 0)
    m.train()
    loss = F.cross_entropy(m(x.to(dev), f.to(dev)), y.to(dev))
    loss.backward()
    keys = [k for k in base if k.split(".")[0] in ("numerical_mlp", "lstm", "classifier")]
    sd = o.unique_params(base, keys)
    ref_loss = F.cross_entropy(o.cnn_lstm_forward(sd, x, f, train=True, dropout_p=0.0), y)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-3 * max(1.0, abs(ref_loss.item()))
    params = dict(m.named_parameters())
    checked = 0
    for k in keys:
        g = params[k].grad
        assert g is not None, k
        assert rel_err(g.cpu(), sd[k].grad) <= 2e-3, (k, rel_err(g.cpu(), sd[k].grad))
        checked += 1
    assert checked == 16
    assert all(p.grad is None for p in m.cnn_backbone.parameters())


def test_cnn_lstm_config5_size_bf16_properties():
    """bf16 throughput build at 6 x 16: eval logits within the stated bf16 bound of the oracle, sequences are
    independent in eval mode (running them 6 at a time equals 2 + 4), gradients finite."""
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    B, T = 6, 16
    x = synth.synth_images(B * T, salt=161).view(B, T, 3, 224, 224).to(dev)
    f = synth.synth_pose_features(B * T, salt=161).view(B, T, 47).to(dev)
    y = synth.synth_labels(B, 12, salt=161).to(dev)
    m = _build(torch.bfloat16, T).to(dev).eval()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        all6 = m(x, f).clone()
        parts = torch.cat([m(x[:2], f[:2]).clone(), m(x[2:], f[2:]).clone()])
        ref = o.cnn_lstm_forward(o.cnn_lstm_sd_to_base(sd0), x.cpu(), f.cpu())
    assert torch.equal(all6, parts)
    assert rel_err(all6.cpu(), ref) <= LOGIT_TOL[torch.bfloat16]
    m.train()
    F.cross_entropy(m(x, f), y).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
