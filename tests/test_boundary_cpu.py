"""CPU: the drop-in boundary (module tree, state_dict layout, factories, error
behaviour) and the C-ABI library (loads, exports every symbol of include/qtcnn.h).
No compute is issued here: there is no GPU in this container."""
import ctypes
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

from _util import PKG, ROOT, pkg


def test_state_dict_layout_matches_reference(golden_eval):
    P = pkg()
    m = P.QuadtreeCNN(12)
    assert list(m.state_dict().keys()) == list(golden_eval["meta/state_dict_keys_quadtree"])
    assert [n for n, _ in m.named_parameters()] == list(golden_eval["meta/param_names_quadtree"])
    assert len(m.state_dict()) == 252 and len(list(m.parameters())) == 72
    s = P.StandardResNetCNN(12)
    assert list(s.state_dict().keys()) == list(golden_eval["meta/state_dict_keys_standard"])
    # aliased keys share storage (triple alias of the backbone)
    sd = m.state_dict()
    assert sd["base_cnn.conv1.weight"].data_ptr() == sd["features_extractor.0.weight"].data_ptr()
    assert sd["base_cnn.layer4.0.conv1.weight"].data_ptr() == sd["global_processor.0.0.conv1.weight"].data_ptr()
    assert sum(p.numel() for p in m.parameters()) == 26_499_028


def test_attention_model_state_dict_matches_reference(golden_attn):
    """AttentionHierarchicalCNN (reference models.py:6-101): 134 keys, no base_cnn.* (the ResNet is a local)."""
    P = pkg()
    m = P.AttentionHierarchicalCNN(12)
    assert list(m.state_dict().keys()) == list(golden_attn["meta/state_dict_keys"])
    assert [n for n, _ in m.named_parameters()] == list(golden_attn["meta/param_names"])
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == int(golden_attn["meta/trainable"])
    assert not hasattr(m, "base_cnn") and len(m.state_dict()) == 134
    with pytest.raises(P.QtError):
        m(torch.zeros(1, 3, 224, 224), torch.zeros(1, 47))
    qs = _load_dropin("quadtree_from_scratch")
    assert isinstance(qs.get_model("attention_hierarchical", 12, "cpu", print_num_params=False), P.AttentionHierarchicalCNN)


def test_cnn_lstm_state_dict_matches_reference(golden_cnn_lstm):
    """CnnLstm (reference cnn+lstm/models.py:14-89): cnn_backbone.* / numerical_mlp / lstm.* / classifier keys."""
    P = pkg()
    m = P.CnnLstm(12, sequence_length=3)
    assert list(m.state_dict().keys()) == list(golden_cnn_lstm["meta/state_dict_keys"])
    assert [n for n, _ in m.named_parameters()] == list(golden_cnn_lstm["meta/param_names"])
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == int(golden_cnn_lstm["meta/trainable"])
    assert not any(p.requires_grad for p in m.cnn_backbone.parameters())
    with pytest.raises(P.QtError):
        m(torch.zeros(1, 3, 3, 224, 224), torch.zeros(1, 3, 47))
    with pytest.raises(ValueError):
        m(torch.zeros(3, 3, 224, 224), torch.zeros(3, 47))
    cl = _load_dropin("cnn_lstm")
    assert isinstance(cl.get_model("cnn_lstm", 12, "cpu", seq_len=4), P.CnnLstm)
    with pytest.raises(ValueError):
        cl.get_model("bogus", 12, "cpu")


def _load_dropin(sub):
    path = os.path.join(ROOT, PKG, sub, "models.py")
    spec = importlib.util.spec_from_file_location(f"dropin_{sub}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dropin_models_surface(golden_eval, capsys):
    qs = _load_dropin("quadtree_from_scratch")
    rn = _load_dropin("resnet")
    m = qs.get_model("quadtree", 12, "cpu")
    assert "Trainable Parameters" in capsys.readouterr().out
    assert all(p.requires_grad for p in m.parameters())
    with pytest.raises(TypeError):  # reference: StandardMultimodalCNN is `pass` (models.py:211-213,318-319)
        qs.get_model("resnet50", 12, "cpu")
    with pytest.raises(RuntimeError):
        qs.get_model("hierarchical_quadtree", 12, "cpu")
    for mode in ("fusion", "image_only", "numerical_only"):
        q = rn.get_model(12, "cpu", mode=mode, print_num_params=False)
        trainable = sum(p.numel() for p in q.parameters() if p.requires_grad)
        assert trainable == int(golden_eval[f"meta/trainable_{mode}"])
        assert not any(p.requires_grad for p in q.base_cnn.parameters())
        assert q.mode == mode and q.gradients is None and q.activations is None
    s = rn.get_model(12, "cpu", mode="standard_resnet_only", print_num_params=False)
    assert type(s).__name__ == "StandardResNetCNN" and len(s.state_dict()) == 246
    with pytest.raises(ValueError):
        rn.QuadtreeCNN(12, mode="bogus")
    # hookable layer4 module and the hook methods of resnet/models.py:135-139
    assert isinstance(q.base_cnn.layer4, torch.nn.Module)
    q.save_activation_hook(None, None, "a")
    q.save_gradient_hook(None, None, ("g",))
    assert q.activations == "a" and q.gradients == "g"


def test_no_cpu_fallback():
    P = pkg()
    m = P.QuadtreeCNN(12)
    with pytest.raises(P.QtError):
        m(torch.zeros(1, 3, 224, 224), torch.zeros(1, 47))
    with pytest.raises(P.QtError):
        m.base_cnn.conv1(torch.zeros(1, 3, 224, 224))


def test_input_validation_happens_before_any_device_work():
    P = pkg()
    m = P.QuadtreeCNN(12)
    with pytest.raises((ValueError, P.QtError)):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 47))


def test_synth_rule_is_deterministic_and_alias_consistent():
    P, synth = pkg(), pkg("synth")
    a = synth.synth_state_dict(P.QuadtreeCNN(12))
    b = synth.synth_state_dict(P.QuadtreeCNN(12))
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["base_cnn.layer1.0.conv1.weight"], a["features_extractor.4.0.conv1.weight"])
    assert float(a["base_cnn.bn1.running_var"].min()) > 0.5


def test_library_exports_every_declared_symbol():
    lib_path = os.path.join(ROOT, PKG, "libqtcnn_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    header = open(os.path.join(ROOT, "include", "qtcnn.h")).read()
    declared = set(re.findall(r"\b(qt_[a-z0-9_]+)\s*\(", header))
    declared -= {"qt_plan_desc", "qt_conv_desc", "qt_dtype", "qt_status"}
    assert len(declared) >= 40
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    lib.qt_version.restype = ctypes.c_int
    assert lib.qt_version() >= 100


def test_plan_tensor_table_matches_module_tree():
    """qt_plan_create touches no GPU: the tensor table must name exactly the model's tensors."""
    P = pkg()
    eng = pkg("engine")
    L = pkg("_lib").lib()
    eng._bind_api(L)
    for kind, model in ((0, P.QuadtreeCNN(12)), (1, P.StandardResNetCNN(12)), (2, P.AttentionHierarchicalCNN(12)),
                        (3, P.CnnLstm(12))):
        desc = eng.PlanDesc(1, 8, 12, kind, 0, 47, 0.5, 1e-5, 0.1, 4, 256)
        h = ctypes.c_void_p()
        assert L.qt_plan_create(ctypes.byref(desc), ctypes.byref(h)) == 0
        tensors = {model._plan_name(n): t for n, t in model.named_parameters()}
        tensors.update({model._plan_name(n): t for n, t in model.named_buffers()})
        dims = (ctypes.c_int * 4)()
        n = L.qt_plan_num_tensors(h)
        names = set()
        for i in range(n):
            name = L.qt_plan_tensor_name(h, i).decode()
            names.add(name)
            nd = L.qt_plan_tensor_shape(h, i, dims)
            assert name in tensors, name
            assert tuple(dims[k] for k in range(nd)) == tuple(tensors[name].shape), name
        assert names == set(tensors.keys())
        assert L.qt_plan_workspace_bytes(h) > 0
        L.qt_plan_destroy(h)
    bad = eng.PlanDesc(5, 8, 12, 0, 0, 47, 0.5, 1e-5, 0.1)
    assert L.qt_plan_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    L.qt_last_error.restype = ctypes.c_char_p
    assert b"dtype" in L.qt_last_error()


def test_gradient_buckets_are_the_surveyed_sizes():
    """The flat gradient buffer of the all-trainable QuadtreeCNN (Quadtree_from scratch variant): 25,986,028 elements =
    103.9 MB f32 per step (base_cnn.fc is never used and has no gradient) in the four phase buckets backward
    finishes them in -- head 59.2 MB, layer4 33.6 MB, layers 3 + 2 10.5 MB, layer1 + stem 0.6 MB (only that one is exposed)."""
    P = pkg()
    eng = pkg("engine")
    L = pkg("_lib").lib()
    eng._bind_api(L)
    model = P.QuadtreeCNN(12)
    desc = eng.PlanDesc(1, 8, 12, 0, 0, 47, 0.5, 1e-5, 0.1, 0, 0)
    h = ctypes.c_void_p()
    assert L.qt_plan_create(ctypes.byref(desc), ctypes.byref(h)) == 0
    names = [L.qt_plan_tensor_name(h, i).decode() for i in range(L.qt_plan_num_tensors(h))]
    L.qt_plan_destroy(h)
    index = {n: i for i, n in enumerate(names)}
    seen, wanted = set(), []
    for n, p in model.named_parameters():
        pn = model._plan_name(n)
        if pn.startswith("base_cnn.fc.") or pn in seen:
            continue
        seen.add(pn)
        wanted.append((index[pn], tuple(p.shape)))
    offs, sizes, ends, total = eng.PlanEngine.gradient_buckets(names, wanted)
    assert sum(sizes.values()) == 25986028
    assert 0 <= total - 25986028 <= 4 * len(wanted)          # (views padded to 16 bytes)
    mb = [4 * (ends[b] - (ends[b - 1] if b else 0)) / 1e6 for b in range(4)]
    assert [round(x, 1) for x in mb] == [59.2, 33.6, 10.5, 0.6], mb
    assert len(set(offs.values())) == len(wanted) and all(o % 4 == 0 for o in offs.values())


def test_conv_descriptor_validation_needs_no_device():
    """qt_conv2d_igemm refuses an inconsistent destination mapping (merged stride-2 data gradient, qtcnn.h) with a
    message and before any HIP call: runs on the CPU box, the pointers are never dereferenced."""
    Lm = pkg("_lib")
    L = Lm.lib()
    L.qt_last_error.restype = ctypes.c_char_p
    d = Lm.ConvDesc()
    d.dtype, d.mode, d.batch = Lm.qt_dtype(torch.bfloat16), Lm.QT_CONV_FWD, 2
    d.in_h = d.in_w = d.out_h = d.out_w = 14
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = 128, 4 * 64, 2, 2, 1, 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = 14 * 14 * 128, 14 * 128, 128
    d.dst_sub, d.dst_h, d.dst_w, d.dst_merge = 2, 28, 28, 64
    fake = ctypes.c_void_p(4096)
    io = Lm.ConvIO(fake, fake, fake, None, None, None, None, None)
    for field, bad in (("n_out", 2 * 64), ("dst_merge", 60), ("dst_sub", 1), ("dst_h", 20)):
        good = getattr(d, field)
        setattr(d, field, bad)
        assert L.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), None) != 0, field
        assert b"dst" in L.qt_last_error(), field
        setattr(d, field, good)


def test_missing_pretrained_weights_are_loud(monkeypatch):
    """The reference always starts from ImageNet weights (resnet18(weights=IMAGENET1K_V1), models.py:221); offline they
    are absent, and a frozen random backbone must not go unnoticed: one UserWarning naming QTCNN_RESNET18_WEIGHTS,
    silenced by pretrained=False or QTCNN_RESNET18_WEIGHTS=none."""
    import warnings
    P, M = pkg(), pkg("modules")
    monkeypatch.delenv("QTCNN_RESNET18_WEIGHTS", raising=False)
    monkeypatch.setattr(M, "_warned_no_weights", False)
    with pytest.warns(UserWarning, match="QTCNN_RESNET18_WEIGHTS"):
        P.StandardResNetCNN(12)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        P.QuadtreeCNN(12)                      # one-time: already warned in this process
        monkeypatch.setattr(M, "_warned_no_weights", False)
        P.QuadtreeCNN(12, pretrained=False)    # explicit opt-out
        monkeypatch.setenv("QTCNN_RESNET18_WEIGHTS", "none")
        P.CnnLstm(12)
    # a real file is loaded
    ref = M.ResNet18()
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "qt_test_resnet18.pth")
    torch.save(ref.state_dict(), path)
    monkeypatch.setenv("QTCNN_RESNET18_WEIGHTS", path)
    m = P.QuadtreeCNN(12)
    assert torch.equal(m.base_cnn.layer3[1].conv2.weight, ref.layer3[1].conv2.weight)
    os.remove(path)


def test_gradients_are_only_requested_where_the_mode_produces_them():
    """resnet/models.py:141-180: the branch a `mode` never runs is not in the autograd graph, its parameters keep
    grad None (Adam skips them, the checkpoint keeps their values).  The plan skips the same branches, so the module
    must not hand it gradient buffers for them."""
    P = pkg()
    cases = {"fusion": lambda n: True,
             "image_only": lambda n: not n.startswith("numerical_mlp."),
             "numerical_only": lambda n: n.startswith(("numerical_mlp.", "classifier."))}
    for mode, produced in cases.items():
        m = P.QuadtreeCNN(12, mode=mode)
        for n, _ in m.named_parameters():
            want = produced(n) and not n.startswith("base_cnn.fc.")
            assert m._plan_writes_grad(m._plan_name(n)) == want, (mode, n)


def test_hooks_on_unserved_submodules_raise_instead_of_staying_dead():
    P = pkg()
    QtError = pkg("_lib").QtError
    m = P.QuadtreeCNN(12)
    h = m.base_cnn.layer4.register_forward_hook(lambda *a: None)     # the Grad-CAM hook point: served
    m.register_forward_hook(lambda *a: None)                          # on the model itself: torch fires it
    m._check_hooks()
    h2 = m.base_cnn.layer3.register_forward_hook(lambda *a: None)
    with pytest.raises(QtError, match="layer3"):
        m._check_hooks()
    h2.remove()
    h3 = m.classifier[0].register_full_backward_hook(lambda *a: None)
    with pytest.raises(QtError, match="classifier.0"):
        m._check_hooks()
    h3.remove(); h.remove()
    m._check_hooks()


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """`--gpus N` under a launcher with another WORLD_SIZE must fail loudly (round-1 bug: it silently benched 1 GPU)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert r.stdout.strip() == ""
