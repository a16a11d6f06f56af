"""CPU: the oracle restatement is pinned by the vectors the reference itself produced
(tests/golden/*.npz, generator tests/golden/make_golden.py)."""
import sys

import numpy as np
import pytest
import torch

from _util import ROOT, check_summary, pkg, rel_err, summary

sys.path.insert(0, ROOT)
import oracle.quadtree_oracle as o  # noqa: E402

TOL = 2e-5  # same torch CPU kernels, different call graph (functional vs nn.Module): rounding only


def _sd(kind="quadtree", mode="fusion"):
    P, synth = pkg(), pkg("synth")
    m = P.StandardResNetCNN(12) if kind == "standard" else P.QuadtreeCNN(12, mode=mode)
    return synth.synth_state_dict(m)


def test_oracle_eval_logits_and_taps(golden_eval):
    torch.set_num_threads(8)
    synth = pkg("synth")
    x, f = synth.synth_images(2, salt=0), synth.synth_pose_features(2, salt=0)
    taps = {}
    with torch.no_grad():
        logits = o.quadtree_forward(_sd(), x, f, taps=taps)
    assert rel_err(logits, golden_eval["qs_quadtree_eval/logits"]) <= TOL
    for name in ("stem", "layer1", "layer2", "layer3", "layer4", "numerical_features", "hidden"):
        check_summary(taps[name], golden_eval, f"qs_quadtree_eval/tap/{name}", TOL)
    for mode in ("fusion", "image_only", "numerical_only"):
        with torch.no_grad():
            logits = o.quadtree_forward(_sd(mode=mode), x, f, mode=mode)
        assert rel_err(logits, golden_eval[f"rn_{mode}_eval/logits"]) <= TOL
    with torch.no_grad():
        logits = o.standard_resnet_forward(_sd("standard"), x)
    assert rel_err(logits, golden_eval["rn_standard_eval/logits"]) <= TOL


def test_oracle_rejects_bad_mode():
    with pytest.raises(ValueError):
        o.quadtree_forward({}, None, None, mode="standard_resnet_only")


@pytest.mark.parametrize("case,frozen", [("qs_quadtree_train", False), ("rn_fusion_train", True)])
def test_oracle_train_step(case, frozen, golden_train):
    torch.set_num_threads(8)
    synth = pkg("synth")
    B = 4
    x, f, y = synth.synth_images(B, salt=1), synth.synth_pose_features(B, salt=1), synth.synth_labels(B, 12, salt=1)
    sd0 = _sd()
    keys = o.trainable_keys(sd0, frozen)
    sd = o.unique_params(sd0, keys)
    logits = o.quadtree_forward(sd, x, f, train=True, dropout_p=0.0)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    assert rel_err(logits.detach(), golden_train[f"{case}/logits"]) <= TOL
    assert abs(loss.item() - float(golden_train[f"{case}/loss"])) <= 1e-4
    names = list(golden_train[f"{case}/grad_names"])
    assert sorted(names) == sorted(keys)
    for n in names:
        pre = f"{case}/grad/{n}"
        smp, gold = summary(sd[n].grad)["sample"], golden_train[f"{pre}/sample"]
        # rounding-level differences may flip an isolated ReLU (see tests/test_model_gpu.py)
        tol = 5e-2 if n.startswith("base_cnn.") else 1e-4
        assert float(np.abs(smp - gold).max()) <= tol * max(float(np.abs(gold).max()), 1e-30), n
    for k in golden_train.files:
        if k.startswith(f"{case}/buf/") and k.endswith("/shape"):
            n = k[len(f"{case}/buf/"):-len("/shape")]
            check_summary(sd[n], golden_train, f"{case}/buf/{n}", 1e-5)


def test_oracle_attention_model(golden_attn):
    """attention_forward against the reference's AttentionHierarchicalCNN (models.py:6-101): eval logits and
    taps, then a dropout-free train step (logits, loss, every gradient, BatchNorm running statistics)."""
    torch.set_num_threads(8)
    P, synth = pkg(), pkg("synth")
    B = 2
    x, f, y = synth.synth_images(B, salt=5), synth.synth_pose_features(B, salt=5), synth.synth_labels(B, 12, salt=5)
    sd_model = synth.synth_state_dict(P.AttentionHierarchicalCNN(12))
    taps = {}
    with torch.no_grad():
        logits = o.attention_forward(o.attention_sd_to_base(sd_model), x, f, taps=taps)
    assert rel_err(logits, golden_attn["eval/logits"]) <= TOL
    for name in ("layer2", "sub_vectors", "attention_weights", "fused"):
        check_summary(taps[name], golden_attn, f"eval/tap/{name}", TOL)
    names = list(golden_attn["train/grad_names"])
    leaves = {n: sd_model[n].clone().requires_grad_(True) for n in names}
    sd = o.attention_sd_to_base({k: leaves.get(k, v.clone()) for k, v in sd_model.items()})
    logits = o.attention_forward(sd, x, f, train=True, dropout_p=0.0)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    assert rel_err(logits.detach(), golden_attn["train/logits"]) <= TOL
    assert abs(loss.item() - float(golden_attn["train/loss"])) <= 1e-4
    for n in names:
        smp, gold = summary(leaves[n].grad)["sample"], golden_attn[f"train/grad/{n}/sample"]
        head = n.split(".")[0] in ("quadrant_processor", "sub_quadrant_processor", "attention_gate", "numerical_mlp", "classifier")
        tol = 1e-4 if head else 5e-2  # an isolated ReLU may flip under another summation order (tests/test_model_gpu.py)
        # attention_gate.2.bias: softmax is shift invariant, the gradient is rounding noise around zero
        scale = max(float(np.abs(gold).max()), 1e-6 if n == "attention_gate.2.bias" else 1e-30)
        assert float(np.abs(smp - gold).max()) <= tol * scale, n
    for k in golden_attn.files:
        if k.startswith("train/buf/") and k.endswith("/shape"):
            n = k[len("train/buf/"):-len("/shape")]
            base = next((b_ + n[len(m_):] for m_, b_ in o._ATTN_PREFIX if n.startswith(m_)), n)
            check_summary(sd[base], golden_attn, f"train/buf/{n}", 1e-5)


def test_oracle_cnn_lstm(golden_cnn_lstm):
    """cnn_lstm_forward (LSTM cell written out gate by gate) against the reference's CnnLstm with torch's nn.LSTM
    (cnn+lstm/models.py:14-89): eval logits and taps, then a dropout-free train step with every gradient."""
    torch.set_num_threads(8)
    P, synth = pkg(), pkg("synth")
    g = golden_cnn_lstm
    B, T = 2, 3
    x = synth.synth_images(B * T, salt=7).view(B, T, 3, 224, 224)
    f = synth.synth_pose_features(B * T, salt=7).view(B, T, 47)
    y = synth.synth_labels(B, 12, salt=7)
    sd_model = synth.synth_state_dict(P.CnnLstm(12, sequence_length=T))
    taps = {}
    with torch.no_grad():
        logits = o.cnn_lstm_forward(o.cnn_lstm_sd_to_base(sd_model), x, f, taps=taps)
    assert rel_err(logits, g["eval/logits"]) <= TOL
    for name in ("fused", "lstm_out"):
        check_summary(taps[name], g, f"eval/tap/{name}", TOL)
    names = list(g["train/grad_names"])
    leaves = {n: sd_model[n].clone().requires_grad_(True) for n in names}
    sd = o.cnn_lstm_sd_to_base({k: leaves.get(k, v.clone()) for k, v in sd_model.items()})
    logits = o.cnn_lstm_forward(sd, x, f, train=True, dropout_p=0.0)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    assert rel_err(logits.detach(), g["train/logits"]) <= TOL
    assert abs(loss.item() - float(g["train/loss"])) <= 1e-4
    assert sorted(names) == sorted(n for n in sd_model if n.split(".")[0] in ("numerical_mlp", "lstm", "classifier"))
    for n in names:
        smp, gold = summary(leaves[n].grad)["sample"], g[f"train/grad/{n}/sample"]
        assert float(np.abs(smp - gold).max()) <= 1e-4 * max(float(np.abs(gold).max()), 1e-30), n
    for k in g.files:
        if k.startswith("train/buf/") and k.endswith("/shape"):
            n = k[len("train/buf/"):-len("/shape")]
            base = next((b_ + n[len(m_):] for m_, b_ in o._LSTM_PREFIX if n.startswith(m_)), n)
            check_summary(sd[base], g, f"train/buf/{n}", 1e-5)


def test_numpy_float64_restatement_agrees_with_reference(golden_eval):
    """The torch-independent float64 restatement (oracle/numpy_ops.py) reproduces the reference's
    own logits: operator definitions, BatchNorm constants, pooling semantics and concat order
    are therefore pinned without going through ATen."""
    import oracle.numpy_ops as nops
    synth = pkg("synth")
    x, f = synth.synth_images(2, salt=0), synth.synth_pose_features(2, salt=0)
    logits = nops.quadtree_forward_eval(_sd(), x, f)
    gold = golden_eval["qs_quadtree_eval/logits"].astype(np.float64)
    # the reference ran in fp32: allow fp32 rounding through 20 layers
    assert float(np.abs(logits - gold).max() / np.abs(gold).max()) <= 2e-5


# ---- 3-D clip models: the oracle's restatement against vectors the reference classes produced ----
CLIP_CASES = [("q3_t8", 2, 8, 112, "quadtree_3d_fusion", 31), ("q3_t5", 2, 5, 64, "quadtree_3d_fusion", 31),
              ("q3_img_t8", 2, 8, 64, "quadtree_3d_image_only", 31), ("ji_t4", 2, 4, 64, None, 32)]


def _is_conv_bias_before_bn(name):
    return name.endswith(".0.bias") and (name.startswith("conv3d_") or name.startswith("visual_stream."))


def _clip_inputs(B, T, HW, salt):
    synth = pkg("synth")
    return (synth.synth_images(B * T, salt=salt, size=HW).view(B, T, 3, HW, HW),
            synth.synth_pose_features(B * T, salt=salt, realistic=True).view(B, T, 47), synth.synth_labels(B, 12, salt=salt))


@pytest.mark.parametrize("tag,B,T,HW,mode,salt", CLIP_CASES)
def test_clip3d_oracle_matches_reference_golden(tag, B, T, HW, mode, salt, golden_clip3d):
    import oracle.quadtree_oracle as o
    P, synth = pkg(), pkg("synth")
    g = golden_clip3d
    m = P.Ji3DCNN(12, sequence_length=T) if mode is None else P.Quadtree3DCNN(12, sequence_length=T, mode=mode)
    sd0 = synth.synth_state_dict(m)
    # same keys, same parameter order, same trainable count as the reference class
    assert list(sd0.keys()) == [str(k) for k in g[f"{tag}/meta/state_dict_keys"]]
    assert [n for n, _ in m.named_parameters()] == [str(k) for k in g[f"{tag}/meta/param_names"]]
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == int(g[f"{tag}/meta/trainable"])
    x, f, y = _clip_inputs(B, T, HW, salt)
    fwd = (lambda sd, **kw: o.ji3d_forward(sd, x, f, **kw)) if mode is None else \
        (lambda sd, **kw: o.quadtree3d_forward(sd, x, f, mode=mode, **kw))
    with torch.no_grad():
        taps = {}
        logits = fwd(sd0, taps=taps)
    assert rel_err(logits, g[f"{tag}/eval/logits"]) <= 2e-5
    for name, t in taps.items():
        if f"{tag}/eval/tap/{name}/shape" in g.files:
            check_summary(t, g, f"{tag}/eval/tap/{name}", 2e-5)
    sd = o.clip_params(sd0)
    out = fwd(sd, train=True, dropout_p=0.0)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    assert rel_err(out.detach(), g[f"{tag}/train/logits"]) <= 2e-5
    assert abs(loss.item() - float(g[f"{tag}/train/loss"])) <= 2e-5 * max(1.0, abs(float(g[f"{tag}/train/loss"])))
    for name in [str(n) for n in g[f"{tag}/train/grad_names"]]:
        if _is_conv_bias_before_bn(name):
            # a bias in front of a train-mode BatchNorm has an exactly zero gradient (the batch mean removes it): what
            # either side holds is rounding noise, so it is bounded against the weight gradient instead of compared
            wmax = float(np.abs(g[f"{tag}/train/grad/{name[:-4]}weight/sample"]).max())
            assert float(sd[name].grad.abs().max()) <= 1e-3 * wmax + 1e-6, name
            continue
        check_summary(sd[name].grad, g, f"{tag}/train/grad/{name}", 5e-4)
    for k in g.files:
        if k.startswith(f"{tag}/train/buf/") and k.endswith("/shape"):
            n = k[len(f"{tag}/train/buf/"):-len("/shape")]
            check_summary(sd[n], g, f"{tag}/train/buf/{n}", 2e-5)
