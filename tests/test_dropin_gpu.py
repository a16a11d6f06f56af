"""The reference's own call sites, run through the real boundary on the GPU: each drop-in directory is put on
sys.path exactly as the reference's scripts find their sibling file (`from models import get_model`,
Quadtree_from scratch/Quadtree_train.py:11, resnet/train_cnn_model.py:11, cnn+lstm/training.py:11), the model is built
with `get_model(..., device)` and the loop body of the training script runs unchanged:

  Quadtree_from scratch/Quadtree_train.py:43-45  get_model / CrossEntropyLoss / optim.Adam(lr, weight_decay)
                                       :60-71  .to(device), zero_grad, forward, loss, backward, step, loss.item(),
                                               torch.max(outputs.data, 1), (predicted == labels).sum().item()
                                       :79-91  model.eval() + torch.no_grad() validation pass
                                       :104    torch.save(model.state_dict(), path)  (+ load into a fresh model)
  resnet/train_cnn_model.py:62-65,80-103       the same with get_model(num_classes, device, mode=TRAINING_MODE)
  cnn+lstm/training.py:34-74,86-93             get_model(MODEL_TYPE, num_classes, device, seq_len=SEQ_LEN), Adam(lr)

and every step is compared with the CPU oracle running the same loop (same synthetic weights, Dropout.p = 0 on both
sides as in the golden fixtures: the reference's RNG stream cannot be reproduced on another device).  The parity build
(QTCNN_DTYPE=f32, exact-f32 MFMA) carries the tolerances; the default bf16 build runs the same loop for finiteness,
decreasing loss and checkpoint round trip.
"""
import importlib
import os
import sys

import pytest
import torch
import torch.nn as nn
import torch.optim as optim

from _util import PKG, ROOT, pkg, rel_err

pytestmark = pytest.mark.gpu
NUM_CLASSES = 12


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _oracle():
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    return o


class _sibling_models:
    """`from models import get_model` with <pkg>/<sub>/ first on sys.path, as the reference's scripts do."""

    def __init__(self, sub):
        self.dir = os.path.join(ROOT, PKG, sub)

    def __enter__(self):
        sys.modules.pop("models", None)
        sys.path.insert(0, self.dir)
        mod = importlib.import_module("models")
        assert os.path.dirname(os.path.abspath(mod.__file__)) == self.dir
        return mod

    def __exit__(self, *exc):
        sys.path.remove(self.dir)
        sys.modules.pop("models", None)


def _no_dropout(model):
    """Dropout.p = 0 on the module tree (what tests/golden/make_golden.py does to the reference) + the plan's rate."""
    for m in model.modules():
        if type(m).__name__ == "Dropout":
            m.p = 0.0
    model.dropout_rate = 0.0
    return model


def _loop_body(model, criterion, optimizer, batches, device):
    """Quadtree_train.py:55-71 / train_cnn_model.py:74-103 / cnn+lstm/training.py:34-54, verbatim."""
    model.train()
    running_loss, correct_train, total_train, losses, outs = 0.0, 0, 0, [], []
    for images, numerical_features, labels in batches:
        images, numerical_features, labels = images.to(device), numerical_features.to(device), labels.to(device)
        optimizer.zero_grad()
        outputs = model(images, numerical_features)
        loss = criterion(outputs, labels)
        loss.backward()
        optimizer.step()
        running_loss += loss.item() * labels.size(0)
        _, predicted = torch.max(outputs.data, 1)
        total_train += labels.size(0)
        correct_train += (predicted == labels).sum().item()
        losses.append(loss.item())
        outs.append(outputs.detach().float().cpu())
    return losses, outs, correct_train, total_train


def _validate(model, criterion, batches, device):
    """Quadtree_train.py:79-91."""
    model.eval()
    val_loss, correct_val, total_val, outs = 0.0, 0, 0, []
    with torch.no_grad():
        for images, numerical_features, labels in batches:
            images, numerical_features, labels = images.to(device), numerical_features.to(device), labels.to(device)
            outputs = model(images, numerical_features)
            loss = criterion(outputs, labels)
            val_loss += loss.item() * labels.size(0)
            _, predicted = torch.max(outputs.data, 1)
            total_val += labels.size(0)
            correct_val += (predicted == labels).sum().item()
            outs.append(outputs.float().cpu())
    return val_loss / total_val, outs


def _batches(n, B, salt, seq=None):
    synth = pkg("synth")
    out = []
    for i in range(n):
        if seq:
            x = synth.synth_images(B * seq, salt=salt + i).view(B, seq, 3, 224, 224)
            f = synth.synth_pose_features(B * seq, salt=salt + i).view(B, seq, 47)
        else:
            x, f = synth.synth_images(B, salt=salt + i), synth.synth_pose_features(B, salt=salt + i)
        out.append((x, f, synth.synth_labels(B, NUM_CLASSES, salt=salt + i)))
    return out


def _oracle_loop(o, forward, sd, keys, batches, adam_kw):
    params = [sd[k] for k in keys]
    opt = optim.Adam(params, **adam_kw)
    losses, outs = [], []
    for x, f, y in batches:
        opt.zero_grad()
        out = forward(sd, x, f, train=True, dropout_p=0.0)
        loss = nn.functional.cross_entropy(out, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        outs.append(out.detach())
    return losses, outs


def _checkpoint_round_trip(make_model, model, val_batches, criterion, device, tmp_path):
    path = os.path.join(tmp_path, "pose_model.pth")
    torch.save(model.state_dict(), path)                     # Quadtree_train.py:104
    fresh = make_model()
    fresh.load_state_dict(torch.load(path, map_location=device))   # grad_cam_analysis.py:358 / evaluate_model_cnn.py:66
    _, a = _validate(model, criterion, val_batches, device)
    _, b = _validate(fresh, criterion, val_batches, device)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_quadtree_from_scratch_train_script_loop(dtype, monkeypatch, tmp_path, capsys):
    device = _dev()
    o = _oracle()
    synth = pkg("synth")
    monkeypatch.setenv("QTCNN_DTYPE", dtype)
    with _sibling_models("quadtree_from_scratch") as models:
        make = lambda: _no_dropout(models.get_model(num_classes=NUM_CLASSES, device=device, model_name="quadtree"))
        model = make()
        assert "Trainable Parameters" in capsys.readouterr().out
        model.load_state_dict({k: v.to(device) for k, v in synth.synth_state_dict(model).items()})
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        criterion = nn.CrossEntropyLoss()
        optimizer = optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)   # Quadtree_train.py:45
        train, val = _batches(2, 5, 400), _batches(1, 3, 410)
        losses, outs, correct, total = _loop_body(model, criterion, optimizer, train, device)
        assert total == 10 and 0 <= correct <= total
        assert model.base_cnn.fc.weight.grad is None and model.base_cnn.conv1.weight.grad is not None
        val_loss, vouts = _validate(model, criterion, val, device)
        assert all(torch.isfinite(t).all() for t in outs + vouts) and val_loss == val_loss
        if dtype == "f32":
            keys = o.trainable_keys(sd0, False)
            sd = o.unique_params(sd0, keys)
            rl, ro = _oracle_loop(o, o.quadtree_forward, sd, keys, train, dict(lr=1e-4, weight_decay=1e-4))
            # step 1: identical weights on both sides -> the 1e-3 logits bar; step 2 runs on Adam-updated weights
            # (Adam normalises rounding-level gradients to +-lr: trajectories separate at the 1e-3 level, see
            # test_three_optimizer_steps_follow_the_oracle_trajectory)
            assert rel_err(outs[0], ro[0]) <= 1e-3
            assert abs(losses[0] - rl[0]) <= 1e-3 * max(1.0, abs(rl[0]))
            assert abs(losses[1] - rl[1]) <= 2e-2 * max(1.0, abs(rl[1]))
            with torch.no_grad():
                ve = o.quadtree_forward(sd, val[0][0], val[0][1])
            assert rel_err(vouts[0], ve) <= 2e-2
        _checkpoint_round_trip(make, model, val, criterion, device, tmp_path)


@pytest.mark.parametrize("mode", ["fusion", "image_only", "numerical_only", "standard_resnet_only"])
def test_resnet_train_script_loop(mode, monkeypatch, tmp_path, capsys):
    device = _dev()
    o = _oracle()
    synth = pkg("synth")
    monkeypatch.setenv("QTCNN_DTYPE", "f32")
    with _sibling_models("resnet") as models:
        make = lambda: _no_dropout(models.get_model(num_classes=NUM_CLASSES, device=device, mode=mode))  # :62
        model = make()
        assert f"(Mode: {mode})" in capsys.readouterr().out
        model.load_state_dict({k: v.to(device) for k, v in synth.synth_state_dict(model).items()})
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        criterion = nn.CrossEntropyLoss()
        optimizer = optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)   # :65: every parameter, frozen ones too
        train, val = _batches(2, 4, 420), _batches(1, 3, 430)
        before = {k: v.detach().clone() for k, v in model.named_parameters()}
        losses, outs, correct, total = _loop_body(model, criterion, optimizer, train, device)
        val_loss, vouts = _validate(model, criterion, val, device)
        std = mode == "standard_resnet_only"
        keys = [k for k in o.trainable_keys(sd0, True)
                if not (mode == "image_only" and k.startswith("numerical_mlp."))
                and not (mode == "numerical_only" and k.startswith("quadrant_processor."))]
        sd = o.unique_params(sd0, keys)
        fwd = (lambda sd_, x, f, **kw: o.standard_resnet_forward(sd_, x, **kw)) if std else \
            (lambda sd_, x, f, **kw: o.quadtree_forward(sd_, x, f, mode=mode, **kw))
        rl, ro = _oracle_loop(o, fwd, sd, keys, train, dict(lr=1e-4, weight_decay=1e-4))
        assert rel_err(outs[0], ro[0]) <= 1e-3
        assert abs(losses[0] - rl[0]) <= 1e-3 * max(1.0, abs(rl[0]))
        assert abs(losses[1] - rl[1]) <= 2e-2 * max(1.0, abs(rl[1]))
        # frozen backbone: no gradient, untouched by Adam; the branch the mode never runs: grad None, untouched
        # (reference: not in the autograd graph, Adam skips grad None -- resnet/models.py:77-78,141-180)
        after = dict(model.named_parameters())
        for k, p in after.items():
            unused = k.startswith("base_cnn.") or k.startswith("features_extractor.") or k.startswith("global_processor.") \
                or (mode == "image_only" and k.startswith("numerical_mlp.")) \
                or (mode == "numerical_only" and k.startswith("quadrant_processor."))
            if unused:
                assert p.grad is None, k
                assert torch.equal(p.detach(), before[k]), k
        for k in ("classifier.0.weight", "classifier.3.bias"):
            assert not torch.equal(after[k].detach(), before[k]), k
        # train-mode BatchNorm of the frozen backbone still moved its running statistics (resnet/train_cnn_model.py:74)
        if mode != "numerical_only":
            assert not torch.equal(model.base_cnn.bn1.running_mean.cpu(), sd0["base_cnn.bn1.running_mean"])
        _checkpoint_round_trip(make, model, val, criterion, device, tmp_path)


def test_cnn_lstm_training_script_loop(monkeypatch, tmp_path):
    device = _dev()
    o = _oracle()
    synth = pkg("synth")
    monkeypatch.setenv("QTCNN_DTYPE", "f32")
    SEQ_LEN = 4   # cnn+lstm/training.py:21
    with _sibling_models("cnn_lstm") as models:
        make = lambda: _no_dropout(models.get_model("cnn_lstm", NUM_CLASSES, device, seq_len=SEQ_LEN))   # :86
        model = make()
        model.lstm.dropout = 0.0
        model.load_state_dict({k: v.to(device) for k, v in synth.synth_state_dict(model).items()})
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        criterion = nn.CrossEntropyLoss()
        optimizer = optim.Adam(model.parameters(), lr=1e-4)   # :93
        train, val = _batches(2, 2, 440, seq=SEQ_LEN), _batches(1, 2, 450, seq=SEQ_LEN)
        losses, outs, correct, total = _loop_body(model, criterion, optimizer, train, device)
        val_loss, vouts = _validate(model, criterion, val, device)
        base = o.cnn_lstm_sd_to_base(sd0)
        keys = [k for k in base if k.split(".")[0] in ("numerical_mlp", "lstm", "classifier")]
        sd = o.unique_params(base, keys)
        rl, ro = _oracle_loop(o, o.cnn_lstm_forward, sd, keys, train, dict(lr=1e-4))
        assert rel_err(outs[0], ro[0]) <= 1e-3
        assert abs(losses[0] - rl[0]) <= 1e-3 * max(1.0, abs(rl[0]))
        assert abs(losses[1] - rl[1]) <= 2e-2 * max(1.0, abs(rl[1]))
        assert all(p.grad is None for p in model.cnn_backbone.parameters())
        _checkpoint_round_trip(make, model, val, criterion, device, tmp_path)
        with pytest.raises(ValueError):
            models.get_model("bogus", NUM_CLASSES, device)
