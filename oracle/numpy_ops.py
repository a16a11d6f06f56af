"""Second, torch-independent restatement of the QuadtreeCNN eval forward in NumPy float64
(TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product).

Purpose: the primary oracle (oracle/quadtree_oracle.py) calls the same ATen kernels the
reference calls; this file re-derives every operator from its definition (cross-correlation
with zero padding, BatchNorm with eps 1e-5, max-pool with -inf padding and floor mode,
adaptive average pool = mean, Linear = x W^T + b, the reference's concat order) so that a
mistake shared by both torch-based sides would still show.  float64 also makes it the accuracy
referee between the f32 CPU path and the f32-MFMA GPU path.

Follows: /root/reference/Quadtree_from scratch/models.py:222-243,273-305 and torchvision's
ResNet-18 BasicBlock wiring.  Layout: NHWC arrays, OIHW weights as in the
reference state_dict.
"""
import numpy as np

EPS = 1e-5


def conv2d(x, w, b=None, stride=1, pad=0):
    """x [B,H,W,C], w [O,I,kh,kw] -> [B,Ho,Wo,O]; out[b,i,j,o] = sum x[b,i*s-p+u,j*s-p+v,c] w[o,c,u,v]."""
    B, H, W, C = x.shape
    O, I, KH, KW = w.shape
    assert I == C
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, C), dtype=np.float64)
    xp[:, pad:pad + H, pad:pad + W] = x
    out = np.zeros((B, Ho, Wo, O), dtype=np.float64)
    for u in range(KH):
        for v in range(KW):
            patch = xp[:, u:u + (Ho - 1) * stride + 1:stride, v:v + (Wo - 1) * stride + 1:stride]
            out += patch @ w[:, :, u, v].T.astype(np.float64)
    if b is not None:
        out += b
    return out


def batchnorm_eval(x, gamma, beta, mean, var):
    return (x - mean) / np.sqrt(var + EPS) * gamma + beta


def relu(x):
    return np.maximum(x, 0.0)


def maxpool(x, k, stride, pad):
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xp = np.full((B, H + 2 * pad, W + 2 * pad, C), -np.inf)
    xp[:, pad:pad + H, pad:pad + W] = x
    out = np.full((B, Ho, Wo, C), -np.inf)
    for u in range(k):
        for v in range(k):
            out = np.maximum(out, xp[:, u:u + (Ho - 1) * stride + 1:stride, v:v + (Wo - 1) * stride + 1:stride])
    return out


def linear(x, w, b):
    return x @ w.T.astype(np.float64) + b


def _f(sd, k):
    return sd[k].detach().cpu().numpy().astype(np.float64)


def _bn(sd, prefix, x):
    return batchnorm_eval(x, _f(sd, prefix + ".weight"), _f(sd, prefix + ".bias"),
                          _f(sd, prefix + ".running_mean"), _f(sd, prefix + ".running_var"))


def _block(sd, prefix, x, stride):
    out = relu(_bn(sd, prefix + ".bn1", conv2d(x, _f(sd, prefix + ".conv1.weight"), None, stride, 1)))
    out = _bn(sd, prefix + ".bn2", conv2d(out, _f(sd, prefix + ".conv2.weight"), None, 1, 1))
    if prefix + ".downsample.0.weight" in sd:
        x = _bn(sd, prefix + ".downsample.1", conv2d(x, _f(sd, prefix + ".downsample.0.weight"), None, stride, 0))
    return relu(out + x)


def quadtree_forward_eval(sd, image_nchw, numerical):
    """float64 logits [B,C] of QuadtreeCNN (fusion mode) in eval()."""
    x = np.transpose(image_nchw.detach().cpu().numpy().astype(np.float64), (0, 2, 3, 1))
    x = relu(_bn(sd, "base_cnn.bn1", conv2d(x, _f(sd, "base_cnn.conv1.weight"), None, 2, 3)))
    x = maxpool(x, 3, 2, 1)
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _block(sd, f"base_cnn.{name}.0", x, stride)
        x = _block(sd, f"base_cnn.{name}.1", x, 1)
    base = x                                             # [B,14,14,256]
    g = _block(sd, "base_cnn.layer4.0", base, 2)
    g = _block(sd, "base_cnn.layer4.1", g, 1).mean(axis=(1, 2))     # AdaptiveAvgPool2d(1,1) + flatten
    h, w = base.shape[1] // 2, base.shape[2] // 2
    feats = [g]
    for q in (base[:, :h, :w], base[:, :h, w:], base[:, h:, :w], base[:, h:, w:]):
        y = relu(conv2d(q, _f(sd, "quadrant_processor.0.weight"), _f(sd, "quadrant_processor.0.bias"), 1, 1))
        y = maxpool(y, 2, 2, 0)                          # 7 -> 3 (floor)
        feats.append(np.transpose(y, (0, 3, 1, 2)).reshape(y.shape[0], -1))   # flatten(1) of NCHW: c*9+h*3+w
    z = relu(linear(numerical.detach().cpu().numpy().astype(np.float64),
                    _f(sd, "numerical_mlp.0.weight"), _f(sd, "numerical_mlp.0.bias")))
    z = linear(z, _f(sd, "numerical_mlp.3.weight"), _f(sd, "numerical_mlp.3.bias"))
    fused = np.concatenate(feats + [z], axis=1)
    hid = relu(linear(fused, _f(sd, "classifier.0.weight"), _f(sd, "classifier.0.bias")))
    return linear(hid, _f(sd, "classifier.3.weight"), _f(sd, "classifier.3.bias"))
