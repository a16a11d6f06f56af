"""CPU oracle for the QuadtreeCNN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional PyTorch-CPU fp32 restatement of the reference graph, driven by a
flat `state_dict` (the reference's own key layout).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product package never does and fails loudly when its HIP
library is missing.

Pinned by: tests/golden/*.npz, produced by tests/golden/make_golden.py which
imports the reference's own `models.py` files by path and runs them on CPU
(tests/test_oracle_golden.py checks this restatement against those vectors).
The arithmetic itself lives in torch.nn / ATen and in torchvision's ResNet-18
topology (both unpinned by the reference: no requirements file, no tests) --
so parity is pinned to torch 2.10.0 CPU fp32 kernels, not by the reference.

What each function follows:
  resnet18 stem/stages  torchvision ResNet-18; reached by the
                        reference at Quadtree_from scratch/models.py:221-230,
                        resnet/models.py:76-88
  quadrant split        Quadtree_from scratch/models.py:277-282, resnet/models.py:151-156
  quadrant head         Quadtree_from scratch/models.py:234-238,284-287
  global branch         Quadtree_from scratch/models.py:240-243,289; resnet/models.py:148-149
  concat order          Quadtree_from scratch/models.py:291-294,300
  numerical MLP         Quadtree_from scratch/models.py:255-260,297
  classifier            Quadtree_from scratch/models.py:266-271,303; resnet/models.py:115-129
  StandardResNetCNN     resnet/models.py:7-65
  AttentionHierarchicalCNN  Quadtree_from scratch/models.py:6-101 (attention_forward; its state_dict has no
                        base_cnn.* keys, see attention_sd_to_base)
  Quadtree3DCNN         3dcnn/models.py:96-214 (quadtree3d_forward), Ji3DCNN cnn+lstm/models.py:93-142 (ji3d_forward):
                        Conv3d / BatchNorm3d / MaxPool3d through torch's CPU kernels, LSTM written out gate by gate
  CnnLstm               cnn+lstm/models.py:14-89 (cnn_lstm_forward; keys via cnn_lstm_sd_to_base); the LSTM
                        cell is written out gate by gate (torch.nn.LSTM semantics: gate rows i,f,g,o,
                        c' = f*c + i*g, h' = o*tanh(c'), dropout on layer 0's outputs as layer 1's input)
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn(sd, prefix, x, train):
    return F.batch_norm(
        x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
        sd[prefix + ".weight"], sd[prefix + ".bias"],
        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _basic_block(sd, prefix, x, stride, train):
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(sd, prefix + ".bn1", out, train))
    out = F.conv2d(out, sd[prefix + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, prefix + ".bn2", out, train)
    if (prefix + ".downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0)
        idn = _bn(sd, prefix + ".downsample.1", idn, train)
    else:
        idn = x
    return F.relu(out + idn)


def _layer(sd, name, x, stride, train):
    x = _basic_block(sd, f"base_cnn.{name}.0", x, stride, train)
    return _basic_block(sd, f"base_cnn.{name}.1", x, 1, train)


def features_to_layer3(sd, image, train=False, taps=None):
    """conv1-bn1-relu-maxpool-layer1-layer2-layer3: [B,3,224,224] -> [B,256,14,14]."""
    x = F.conv2d(image, sd["base_cnn.conv1.weight"], None, 2, 3)
    x = F.relu(_bn(sd, "base_cnn.bn1", x, train))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["stem"] = x
    x = _layer(sd, "layer1", x, 1, train)
    if taps is not None:
        taps["layer1"] = x
    x = _layer(sd, "layer2", x, 2, train)
    if taps is not None:
        taps["layer2"] = x
    x = _layer(sd, "layer3", x, 2, train)
    if taps is not None:
        taps["layer3"] = x
    return x


def _dropout(x, p, train, masks, name):
    """Dropout with an optional injected keep-mask (values in {0,1}); the torch
    RNG stream of the reference cannot be reproduced on another device, so
    train-mode parity is checked with p=0 or an injected mask."""
    if not train or p == 0.0:
        return x
    if masks is not None and name in masks:
        return x * masks[name] / (1.0 - p)
    return F.dropout(x, p, True)


def quadrant_head(sd, q):
    y = F.conv2d(q, sd["quadrant_processor.0.weight"], sd["quadrant_processor.0.bias"], 1, 1)
    return F.max_pool2d(F.relu(y), 2, 2).flatten(1)


def quadtree_forward(sd, image, numerical, mode="fusion", train=False,
                     dropout_p=0.5, masks=None, taps=None):
    """logits[B,C] of QuadtreeCNN.  mode per resnet/models.py:115-122."""
    if mode not in ("fusion", "image_only", "numerical_only"):
        raise ValueError(f"Invalid mode: {mode}")
    feats = []
    if mode in ("fusion", "image_only"):
        base = features_to_layer3(sd, image, train, taps)
        g = _layer(sd, "layer4", base, 2, train)
        if taps is not None:
            taps["layer4"] = g
        g = F.adaptive_avg_pool2d(g, (1, 1)).flatten(1)
        h, w = base.shape[2] // 2, base.shape[3] // 2
        quads = [base[:, :, :h, :w], base[:, :, :h, w:], base[:, :, h:, :w], base[:, :, h:, w:]]
        img = torch.cat([g] + [quadrant_head(sd, q) for q in quads], dim=1)
        if taps is not None:
            taps["image_features"] = img
        feats.append(img)
    if mode in ("fusion", "numerical_only"):
        z = F.relu(F.linear(numerical, sd["numerical_mlp.0.weight"], sd["numerical_mlp.0.bias"]))
        z = _dropout(z, dropout_p, train, masks, "numerical_mlp")
        z = F.linear(z, sd["numerical_mlp.3.weight"], sd["numerical_mlp.3.bias"])
        if taps is not None:
            taps["numerical_features"] = z
        feats.append(z)
    fused = torch.cat(feats, dim=1) if len(feats) > 1 else feats[0]
    hdn = F.relu(F.linear(fused, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    if taps is not None:
        taps["hidden"] = hdn
    hdn = _dropout(hdn, dropout_p, train, masks, "classifier")
    return F.linear(hdn, sd["classifier.3.weight"], sd["classifier.3.bias"])


def standard_resnet_forward(sd, image, train=False, dropout_p=0.5, masks=None, taps=None):
    """logits[B,C] of StandardResNetCNN (numerical input ignored, resnet/models.py:56)."""
    base = features_to_layer3(sd, image, train, taps)
    g = _layer(sd, "layer4", base, 2, train)
    if taps is not None:
        taps["layer4"] = g
    g = F.adaptive_avg_pool2d(g, (1, 1)).flatten(1)
    hdn = F.relu(F.linear(g, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    hdn = _dropout(hdn, dropout_p, train, masks, "classifier")
    return F.linear(hdn, sd["classifier.3.weight"], sd["classifier.3.bias"])


_ATTN_PREFIX = (("features_extractor.0.", "base_cnn.conv1."), ("features_extractor.1.", "base_cnn.bn1."),
                ("features_extractor.4.", "base_cnn.layer1."), ("features_extractor.5.", "base_cnn.layer2."),
                ("global_processor.0.", "base_cnn.layer3."), ("global_processor.1.", "base_cnn.layer4."))


def attention_sd_to_base(sd):
    """AttentionHierarchicalCNN keeps its ResNet-18 as a local of __init__ (models.py:11), so its state_dict
    names the layers features_extractor.{0,1,4,5}.* / global_processor.{0,1}.*; rename them to the base_cnn.*
    names the shared ResNet helpers of this file use.  Tensors are shared, not copied."""
    out = {}
    for k, v in sd.items():
        for mine, base in _ATTN_PREFIX:
            if k.startswith(mine):
                k = base + k[len(mine):]
                break
        out[k] = v
    return out


def attention_forward(sd, image, numerical, train=False, dropout_p=0.5, masks=None, taps=None):
    """logits[B,C] of AttentionHierarchicalCNN (Quadtree_from scratch/models.py:57-101); `sd` uses base_cnn.*
    names (attention_sd_to_base)."""
    x = F.conv2d(image, sd["base_cnn.conv1.weight"], None, 2, 3)
    x = F.max_pool2d(F.relu(_bn(sd, "base_cnn.bn1", x, train)), 3, 2, 1)
    x = _layer(sd, "layer1", x, 1, train)
    base = _layer(sd, "layer2", x, 2, train)                                  # :58  [B,128,28,28]
    g = _layer(sd, "layer4", _layer(sd, "layer3", base, 2, train), 2, train)  # :61
    g = F.adaptive_avg_pool2d(g, (1, 1)).flatten(1)

    def split4(t):                                                            # :63-68, :72-77
        h, w = t.shape[2] // 2, t.shape[3] // 2
        return [t[:, :, :h, :w], t[:, :, :h, w:], t[:, :, h:, :w], t[:, :, h:, w:]]

    def head(t, name):                                                        # :21-30
        y = F.relu(F.conv2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], 1, 1))
        return F.adaptive_avg_pool2d(y, (1, 1)).flatten(1)

    quads = split4(base)
    qf = [head(q, "quadrant_processor") for q in quads]                       # :69
    subs = [head(sq, "sub_quadrant_processor") for q in quads for sq in split4(q)]  # :72-78
    stacked = torch.stack(subs, dim=1)                                        # :81  [B,16,64]
    a = F.relu(F.linear(stacked, sd["attention_gate.0.weight"], sd["attention_gate.0.bias"]))
    scores = F.linear(a, sd["attention_gate.2.weight"], sd["attention_gate.2.bias"]).squeeze(-1)  # :85
    weights = F.softmax(scores, dim=1).unsqueeze(-1)                          # :87
    attended = torch.sum(stacked * weights, dim=1)                            # :89
    img = torch.cat([g] + qf + [attended], dim=1)                             # :92-93
    z = F.relu(F.linear(numerical, sd["numerical_mlp.0.weight"], sd["numerical_mlp.0.bias"]))
    z = _dropout(z, dropout_p, train, masks, "numerical_mlp")                 # :96
    fused = torch.cat((img, z), dim=1)
    if taps is not None:
        taps.update(layer2=base, sub_vectors=stacked, attention_weights=weights.squeeze(-1), fused=fused)
    hdn = F.relu(F.linear(fused, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    hdn = _dropout(hdn, dropout_p, train, masks, "classifier")
    return F.linear(hdn, sd["classifier.3.weight"], sd["classifier.3.bias"])


_LSTM_PREFIX = (("cnn_backbone.0.", "base_cnn.conv1."), ("cnn_backbone.1.", "base_cnn.bn1."),
                ("cnn_backbone.4.", "base_cnn.layer1."), ("cnn_backbone.5.", "base_cnn.layer2."),
                ("cnn_backbone.6.", "base_cnn.layer3."), ("cnn_backbone.7.", "base_cnn.layer4."))


def cnn_lstm_sd_to_base(sd):
    """CnnLstm names its ResNet-18 layers cnn_backbone.{0,1,4,5,6,7}.* (nn.Sequential of resnet.children()[:-1],
    cnn+lstm/models.py:23); rename to the base_cnn.* names used by the shared helpers."""
    out = {}
    for k, v in sd.items():
        for mine, base in _LSTM_PREFIX:
            if k.startswith(mine):
                k = base + k[len(mine):]
                break
        out[k] = v
    return out


def _lstm_layer(sd, layer, x):
    """one nn.LSTM layer, batch_first, zero initial state: x [B,T,I] -> [B,T,H]"""
    w_ih, w_hh = sd[f"lstm.weight_ih_l{layer}"], sd[f"lstm.weight_hh_l{layer}"]
    b_ih, b_hh = sd[f"lstm.bias_ih_l{layer}"], sd[f"lstm.bias_hh_l{layer}"]
    B, T, H = x.shape[0], x.shape[1], w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = []
    for t in range(T):
        gates = F.linear(x[:, t], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, g, o = gates.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def cnn_lstm_forward(sd, image_sequence, numerical_sequence, train=False, dropout_p=0.5, masks=None, taps=None):
    """logits[B,C] of CnnLstm (cnn+lstm/models.py:58-89); `sd` uses base_cnn.* names (cnn_lstm_sd_to_base).
    train=True: the frozen backbone's BatchNorms still use (and update) batch statistics, as model.train() does."""
    B, T = image_sequence.shape[0], image_sequence.shape[1]
    frames = image_sequence.reshape(B * T, *image_sequence.shape[2:])                     # :65
    base = features_to_layer3(sd, frames, train)
    g = _layer(sd, "layer4", base, 2, train)
    c_out = F.adaptive_avg_pool2d(g, (1, 1)).flatten(1).view(B, T, -1)                   # :68-69
    z = F.relu(F.linear(numerical_sequence, sd["numerical_mlp.0.weight"], sd["numerical_mlp.0.bias"]))
    n_out = F.linear(z, sd["numerical_mlp.2.weight"], sd["numerical_mlp.2.bias"])        # :73
    fused = torch.cat((c_out, n_out), dim=2)                                              # :77
    h0 = _lstm_layer(sd, 0, fused)
    h1 = _lstm_layer(sd, 1, _dropout(h0, dropout_p, train, masks, "lstm"))                # :81 (nn.LSTM dropout=)
    final = h1[:, -1, :]                                                                  # :84
    if taps is not None:
        taps.update(fused=fused, lstm_out=h1)
    hdn = F.relu(F.linear(final, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    hdn = _dropout(hdn, dropout_p, train, masks, "classifier")
    return F.linear(hdn, sd["classifier.3.weight"], sd["classifier.3.bias"])


def _lstm_named(sd, prefix, layer, x):
    """_lstm_layer with another parameter prefix (numerical_lstm.*)"""
    view = {f"lstm.{n}_l{layer}": sd[f"{prefix}.{n}_l{layer}"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    return _lstm_layer(view, layer, x)


def _conv3d_block(sd, prefix, x, train, pool):
    """Conv3d(3x3x3, padding 1, bias) + BatchNorm3d + ReLU (+ MaxPool3d(pool)): `prefix`.0 = conv, `prefix`.1 = bn"""
    x = F.conv3d(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"], stride=1, padding=1)
    x = F.batch_norm(x, sd[f"{prefix}.1.running_mean"], sd[f"{prefix}.1.running_var"], sd[f"{prefix}.1.weight"],
                     sd[f"{prefix}.1.bias"], training=train, momentum=0.1, eps=1e-5)
    x = F.relu(x)
    return F.max_pool3d(x, pool, pool) if pool else x


def quadtree3d_forward(sd, image_sequence, numerical_sequence, mode="quadtree_3d_fusion", train=False, dropout_p=0.6,
                       masks=None, taps=None):
    """logits[B,C] of Quadtree3DCNN (3dcnn/models.py:184-214): image_sequence [B,T,3,H,W], numerical_sequence [B,T,47]."""
    x = image_sequence.permute(0, 2, 1, 3, 4)                                                   # :189
    for prefix, pool in (("conv3d_block1", (1, 2, 2)), ("conv3d_block2", (2, 2, 2)), ("conv3d_block3", (2, 2, 2)),
                         ("conv3d_block4_new", (1, 2, 2)), ("conv3d_final_features", None)):     # :191-196
        x = _conv3d_block(sd, prefix, x, train, pool)
        if taps is not None:
            taps[prefix] = x
    feats = F.adaptive_avg_pool3d(x, (1, 1, 1)).flatten(1)                                        # :198
    if mode == "quadtree_3d_fusion":
        h0 = _lstm_named(sd, "numerical_lstm", 0, numerical_sequence)                             # :201
        h1 = _lstm_named(sd, "numerical_lstm", 1, _dropout(h0, dropout_p, train, masks, "lstm"))
        z = F.relu(F.linear(h1[:, -1, :], sd["numerical_projection.0.weight"], sd["numerical_projection.0.bias"]))
        z = _dropout(z, dropout_p, train, masks, "projection")                                    # :203
        feats = torch.cat((feats, z), dim=1)                                                      # :206
    elif mode != "quadtree_3d_image_only":
        raise ValueError(f"Invalid mode during forward pass: {mode}")
    if taps is not None:
        taps["fused"] = feats
    hdn = F.relu(F.linear(feats, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    hdn = _dropout(hdn, dropout_p, train, masks, "classifier")
    return F.linear(hdn, sd["classifier.3.weight"], sd["classifier.3.bias"])                     # :212


def ji3d_forward(sd, image_sequence, numerical_sequence, train=False, dropout_p=0.5, masks=None, taps=None):
    """logits[B,C] of Ji3DCNN (cnn+lstm/models.py:126-142)."""
    x = image_sequence.permute(0, 2, 1, 3, 4)                                                    # :131
    for prefix, pool in (("visual_stream.0", (1, 2, 2)), ("visual_stream.2", (2, 2, 2)), ("visual_stream.4", None)):
        x = _conv3d_block(sd, prefix, x, train, None)      # (the pools are modules of their own here: :101,103)
        if taps is not None:
            taps[prefix] = x
        if pool:
            x = F.max_pool3d(x, pool, pool)
    v_out = F.adaptive_avg_pool3d(x, (1, 1, 1)).flatten(1)                                        # :132
    n_out = _lstm_named(sd, "numerical_lstm", 0, numerical_sequence)[:, -1, :]                    # :135-136
    fused = torch.cat((v_out, n_out), dim=1)                                                      # :139
    hdn = F.relu(F.linear(fused, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    hdn = _dropout(hdn, dropout_p, train, masks, "classifier")
    return F.linear(hdn, sd["classifier.3.weight"], sd["classifier.3.bias"])


def clip_params(sd):
    """requires_grad leaf copies of every parameter of a clip model's state_dict (buffers cloned plain)"""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


def unique_params(sd, keys):
    """Leaf copies (requires_grad) for `keys`; other entries cloned plain.
    Aliased reference keys (features_extractor.*, global_processor.*) are not
    needed by the functional form and are dropped."""
    out = {}
    for k, v in sd.items():
        if not (k.startswith("base_cnn.") or k.startswith("quadrant_processor.")
                or k.startswith("sub_quadrant_processor.") or k.startswith("attention_gate.") or k.startswith("lstm.")
                or k.startswith("numerical_mlp.") or k.startswith("classifier.")):
            continue
        t = v.detach().clone()
        if k in keys and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    return out


def trainable_keys(sd, frozen_backbone):
    keys = []
    for k, v in sd.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("running_mean", "running_var", "num_batches_tracked"):
            continue
        if k.startswith("base_cnn."):
            if frozen_backbone or k.startswith("base_cnn.fc."):
                continue
            keys.append(k)
        elif k.split(".")[0] in ("quadrant_processor", "sub_quadrant_processor", "attention_gate", "numerical_mlp",
                                 "classifier"):
            keys.append(k)
    return keys
