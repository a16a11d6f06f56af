#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE on CPU (build container only).

Imports the reference's own model files by path
  /root/reference/Quadtree_from scratch/models.py
  /root/reference/resnet/models.py
with oracle/torchvision_standin first on sys.path (torchvision is absent here;
the stand-in restates the ResNet-18 topology and never fetches weights), fills
every tensor with the deterministic rule of <pkg>/synth.py, runs eval forwards
and dropout-free train-mode forward+backward passes on seeded inputs and writes
small .npz fixtures next to this file.  Only data is written: logits, per-stage
checksums / strided samples, gradient checksums and the small gradients in
full.  The 100 MB of weights and the inputs are regenerated from the rule on
both sides and are not stored.

Usage (from the repo root):  python tests/golden/make_golden.py            (eval_b2.npz, train_b4.npz)
                             python tests/golden/make_golden.py attention  (attn_b2.npz: AttentionHierarchicalCNN,
                                                                            Quadtree_from scratch/models.py:6-101)
                             python tests/golden/make_golden.py cnn_lstm   (cnn_lstm_b2t3.npz: CnnLstm,
                                                                            cnn+lstm/models.py:14-89)
                             python tests/golden/make_golden.py clip3d     (clip3d.npz: Quadtree3DCNN 3dcnn/models.py:96-214
                                                                            and Ji3DCNN cnn+lstm/models.py:93-142)
"""
import importlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
PKG = "multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd"

sys.path.insert(0, os.path.join(ROOT, "oracle", "torchvision_standin"))
sys.path.insert(0, ROOT)
synth = importlib.import_module(PKG + ".synth")


def load_ref(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def summary(t, nsample=256):
    f = t.detach().double().flatten()
    stride = max(1, f.numel() // nsample)
    return {
        "shape": np.array(t.shape, dtype=np.int64),
        "sum": np.float64(f.sum().item()),
        "abssum": np.float64(f.abs().sum().item()),
        "sample": f[::stride][:nsample].numpy().astype(np.float32),
    }


def put(out, prefix, t, full_below=4097):
    s = summary(t)
    for k, v in s.items():
        out[f"{prefix}/{k}"] = v
    if t.numel() < full_below:
        out[f"{prefix}/full"] = t.detach().float().numpy()


def set_dropout_p(model, p):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p


def hook_taps(model, taps):
    hs = []

    def mk(name):
        def fn(_m, _i, o):
            taps[name] = o.detach()
        return fn
    b = model.base_cnn
    hs.append(b.maxpool.register_forward_hook(mk("stem")))
    for n in ("layer1", "layer2", "layer3", "layer4"):
        hs.append(getattr(b, n).register_forward_hook(mk(n)))
    if hasattr(model, "numerical_mlp"):
        hs.append(model.numerical_mlp.register_forward_hook(mk("numerical_features")))
    hs.append(model.classifier[1].register_forward_hook(mk("hidden")))
    return hs


def eval_case(model, images, feats, out, prefix, with_taps=True):
    model.eval()
    taps = {}
    hs = hook_taps(model, taps) if with_taps else []
    with torch.no_grad():
        logits = model(images, feats)
    for h in hs:
        h.remove()
    out[f"{prefix}/logits"] = logits.numpy()
    for k, v in taps.items():
        put(out, f"{prefix}/tap/{k}", v, full_below=0)


def train_case(model, images, feats, labels, out, prefix):
    model.train()
    set_dropout_p(model, 0.0)
    for p in model.parameters():
        p.grad = None
    logits = model(images, feats)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out[f"{prefix}/logits"] = logits.detach().numpy()
    out[f"{prefix}/loss"] = np.float64(loss.item())
    names = []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(name)
        put(out, f"{prefix}/grad/{name}", p.grad)
    out[f"{prefix}/grad_names"] = np.array(names)
    for name, b in model.named_buffers():
        if name.startswith("base_cnn.") and name.endswith(("running_mean", "running_var")):
            put(out, f"{prefix}/buf/{name}", b, full_below=0)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    qs = load_ref(os.path.join(REF, "Quadtree_from scratch", "models.py"), "ref_qs_models")
    rn = load_ref(os.path.join(REF, "resnet", "models.py"), "ref_rn_models")
    C = 12

    # ---- eval forwards, B=2 ------------------------------------------------
    B = 2
    images = synth.synth_images(B, salt=0)
    feats = synth.synth_pose_features(B, salt=0, realistic=True)
    out = {}
    m = qs.QuadtreeCNN(num_classes=C)
    m.load_state_dict(synth.synth_state_dict(m))
    out["meta/state_dict_keys_quadtree"] = np.array(list(m.state_dict().keys()))
    out["meta/param_names_quadtree"] = np.array([n for n, _ in m.named_parameters()])
    eval_case(m, images, feats, out, "qs_quadtree_eval")
    for mode in ("fusion", "image_only", "numerical_only"):
        m = rn.QuadtreeCNN(num_classes=C, mode=mode)
        m.load_state_dict(synth.synth_state_dict(m))
        eval_case(m, images, feats, out, f"rn_{mode}_eval", with_taps=False)
        out[f"meta/trainable_{mode}"] = np.int64(
            sum(p.numel() for p in m.parameters() if p.requires_grad))
    m = rn.StandardResNetCNN(num_classes=C)
    m.load_state_dict(synth.synth_state_dict(m))
    out["meta/state_dict_keys_standard"] = np.array(list(m.state_dict().keys()))
    eval_case(m, images, None, out, "rn_standard_eval", with_taps=False)
    np.savez_compressed(os.path.join(HERE, "eval_b2.npz"), **out)
    print("eval_b2.npz:", len(out), "arrays")

    # ---- train-mode forward+backward, B=4, dropout p=0 ----------------------
    B = 4
    images = synth.synth_images(B, salt=1)
    feats = synth.synth_pose_features(B, salt=1, realistic=True)
    labels = synth.synth_labels(B, C, salt=1)
    out = {}
    m = qs.QuadtreeCNN(num_classes=C)
    m.load_state_dict(synth.synth_state_dict(m))
    train_case(m, images, feats, labels, out, "qs_quadtree_train")
    m = rn.QuadtreeCNN(num_classes=C, mode="fusion")
    m.load_state_dict(synth.synth_state_dict(m))
    train_case(m, images, feats, labels, out, "rn_fusion_train")
    m = rn.StandardResNetCNN(num_classes=C)
    m.load_state_dict(synth.synth_state_dict(m))
    train_case(m, images, None, labels, out, "rn_standard_train")
    np.savez_compressed(os.path.join(HERE, "train_b4.npz"), **out)
    print("train_b4.npz:", len(out), "arrays")


def main_attention():
    """AttentionHierarchicalCNN (reference Quadtree_from scratch/models.py:6-101), eval and dropout-free train step, B=2."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    qs = load_ref(os.path.join(REF, "Quadtree_from scratch", "models.py"), "ref_qs_models")
    C, B = 12, 2
    out = {}
    m = qs.AttentionHierarchicalCNN(num_classes=C)
    m.load_state_dict(synth.synth_state_dict(m))
    out["meta/state_dict_keys"] = np.array(list(m.state_dict().keys()))
    out["meta/param_names"] = np.array([n for n, _ in m.named_parameters()])
    out["meta/trainable"] = np.int64(sum(p.numel() for p in m.parameters() if p.requires_grad))
    images = synth.synth_images(B, salt=5)
    feats = synth.synth_pose_features(B, salt=5, realistic=True)
    labels = synth.synth_labels(B, C, salt=5)
    taps = {}
    def tap_layer2(_m, _i, o):
        taps["layer2"] = o.detach()

    def tap_gate(_m, i, o):
        taps["sub_vectors"] = i[0].detach()
        taps["attention_weights"] = torch.softmax(o.detach().squeeze(-1), dim=1)

    def tap_fused(_m, i, _o):
        taps["fused"] = i[0].detach()

    hs = [m.features_extractor.register_forward_hook(tap_layer2), m.attention_gate.register_forward_hook(tap_gate),
          m.classifier[0].register_forward_hook(tap_fused)]
    m.eval()
    with torch.no_grad():
        logits = m(images, feats)
    for h in hs:
        h.remove()
    out["eval/logits"] = logits.numpy()
    for k, v in taps.items():
        put(out, f"eval/tap/{k}", v)
    m.train()
    set_dropout_p(m, 0.0)
    logits = m(images, feats)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out["train/logits"] = logits.detach().numpy()
    out["train/loss"] = np.float64(loss.item())
    names = []
    for name, p in m.named_parameters():
        if p.grad is not None:
            names.append(name)
            put(out, f"train/grad/{name}", p.grad)
    out["train/grad_names"] = np.array(names)
    for name, b in m.named_buffers():
        if name.endswith(("running_mean", "running_var")):
            put(out, f"train/buf/{name}", b, full_below=0)
    np.savez_compressed(os.path.join(HERE, "attn_b2.npz"), **out)
    print("attn_b2.npz:", len(out), "arrays")


def main_cnn_lstm():
    """CnnLstm (reference cnn+lstm/models.py:14-89): 2 sequences of 3 frames, eval and dropout-free train step."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cl = load_ref(os.path.join(REF, "cnn+lstm", "models.py"), "ref_cl_models")
    C, B, T = 12, 2, 3
    out = {}
    m = cl.CnnLstm(num_classes=C, sequence_length=T)
    m.load_state_dict(synth.synth_state_dict(m))
    out["meta/state_dict_keys"] = np.array(list(m.state_dict().keys()))
    out["meta/param_names"] = np.array([n for n, _ in m.named_parameters()])
    out["meta/trainable"] = np.int64(sum(p.numel() for p in m.parameters() if p.requires_grad))
    images = synth.synth_images(B * T, salt=7).view(B, T, 3, 224, 224)
    feats = synth.synth_pose_features(B * T, salt=7, realistic=True).view(B, T, 47)
    labels = synth.synth_labels(B, C, salt=7)
    taps = {}

    def tap_lstm(_m, i, o):
        taps["fused"] = i[0].detach()
        taps["lstm_out"] = o[0].detach()

    h = m.lstm.register_forward_hook(tap_lstm)
    m.eval()
    with torch.no_grad():
        logits = m(images, feats)
    h.remove()
    out["eval/logits"] = logits.numpy()
    for k, v in taps.items():
        put(out, f"eval/tap/{k}", v)
    m.train()
    set_dropout_p(m, 0.0)
    m.lstm.dropout = 0.0
    logits = m(images, feats)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out["train/logits"] = logits.detach().numpy()
    out["train/loss"] = np.float64(loss.item())
    names = []
    for name, p in m.named_parameters():
        if p.grad is not None:
            names.append(name)
            put(out, f"train/grad/{name}", p.grad)
    out["train/grad_names"] = np.array(names)
    for name, b in m.named_buffers():
        if name.endswith(("running_mean", "running_var")):
            put(out, f"train/buf/{name}", b, full_below=0)
    np.savez_compressed(os.path.join(HERE, "cnn_lstm_b2t3.npz"), **out)
    print("cnn_lstm_b2t3.npz:", len(out), "arrays")


def clip_case(m, images, feats, labels, out, prefix, conv_names):
    """eval logits + taps, then a dropout-free train step (all gradients, running statistics) of a clip model"""
    taps = {}
    hs = [dict(m.named_modules())[n].register_forward_hook(lambda _m, _i, o, n=n: taps.__setitem__(n, o.detach()))
          for n in conv_names]
    m.eval()
    with torch.no_grad():
        logits = m(images, feats)
    for h in hs:
        h.remove()
    out[f"{prefix}/eval/logits"] = logits.numpy()
    for k, v in taps.items():
        put(out, f"{prefix}/eval/tap/{k}", v)
    m.train()
    set_dropout_p(m, 0.0)
    m.numerical_lstm.dropout = 0.0
    logits = m(images, feats)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out[f"{prefix}/train/logits"] = logits.detach().numpy()
    out[f"{prefix}/train/loss"] = np.float64(loss.item())
    names = []
    for name, p in m.named_parameters():
        if p.grad is not None:
            names.append(name)
            put(out, f"{prefix}/train/grad/{name}", p.grad)
    out[f"{prefix}/train/grad_names"] = np.array(names)
    for name, b in m.named_buffers():
        if name.endswith(("running_mean", "running_var")):
            put(out, f"{prefix}/train/buf/{name}", b, full_below=0)
    out[f"{prefix}/meta/state_dict_keys"] = np.array(list(m.state_dict().keys()))
    out[f"{prefix}/meta/param_names"] = np.array([n for n, _ in m.named_parameters()])
    out[f"{prefix}/meta/trainable"] = np.int64(sum(p.numel() for p in m.parameters() if p.requires_grad))


def main_clip3d():
    """3-D clip models run by the reference's own classes: Quadtree3DCNN (3dcnn/models.py:96-214; T = 8 at 112x112 and
    the trainer's SEQUENCE_LENGTH = 5 at 64x64: odd length, frames dropped by the floor-mode pools) and Ji3DCNN
    (cnn+lstm/models.py:93-142; T = 4 at 64x64)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    q3 = load_ref(os.path.join(REF, "3dcnn", "models.py"), "ref_3d_models")
    cl = load_ref(os.path.join(REF, "cnn+lstm", "models.py"), "ref_cl_models")
    C = 12
    out = {}
    blocks = ["conv3d_block1", "conv3d_block2", "conv3d_block3", "conv3d_block4_new", "conv3d_final_features"]
    for tag, B, T, HW, mode in (("q3_t8", 2, 8, 112, "quadtree_3d_fusion"), ("q3_t5", 2, 5, 64, "quadtree_3d_fusion"),
                                ("q3_img_t8", 2, 8, 64, "quadtree_3d_image_only")):
        m = q3.get_model(C, "cpu", mode=mode, sequence_length=T, print_num_params=False)
        m.load_state_dict(synth.synth_state_dict(m))
        images = synth.synth_images(B * T, salt=31, size=HW).view(B, T, 3, HW, HW)
        feats = synth.synth_pose_features(B * T, salt=31, realistic=True).view(B, T, 47)
        clip_case(m, images, feats, synth.synth_labels(B, C, salt=31), out, tag, blocks)
    m = cl.get_model("3d_cnn", C, "cpu", seq_len=4)
    m.load_state_dict(synth.synth_state_dict(m))
    B, T, HW = 2, 4, 64
    images = synth.synth_images(B * T, salt=32, size=HW).view(B, T, 3, HW, HW)
    feats = synth.synth_pose_features(B * T, salt=32, realistic=True).view(B, T, 47)
    clip_case(m, images, feats, synth.synth_labels(B, C, salt=32), out, "ji_t4",
              ["visual_stream.0", "visual_stream.2", "visual_stream.4"])
    np.savez_compressed(os.path.join(HERE, "clip3d.npz"), **out)
    print("clip3d.npz:", len(out), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "clip3d":
        main_clip3d()
    elif len(sys.argv) > 1 and sys.argv[1] == "attention":
        main_attention()
    elif len(sys.argv) > 1 and sys.argv[1] == "cnn_lstm":
        main_cnn_lstm()
    else:
        main()
