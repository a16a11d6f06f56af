"""GPU parity of the 3-D clip models (BASELINE config 4): Quadtree3DCNN
(/root/reference/3dcnn/models.py:96-214) and Ji3DCNN (/root/reference/cnn+lstm/models.py:93-142) on the HIP kernels.

* op level through the C ABI: MaxPool3d (1,2,2) / (2,2,2) forward + backward, BatchNorm3d statistics, the 27-tap clip
  packing, AdaptiveAvgPool3d -- against torch CPU;
* whole models against the vectors the reference classes produced (tests/golden/clip3d.npz): eval logits + block outputs,
  a dropout-free train step (loss, every gradient, running statistics); f32 build: logits <= 1e-3, head / LSTM gradients
  <= 1e-3, conv gradients by the ReLU-flip-aware rule of tests/test_model_gpu.py; bf16 build: stated loose bounds;
* BASELINE config 4 at its own size (T = 8 frames of 224x224) against the CPU oracle run on the box;
* the drop-in `from models import get_model` of 3dcnn/ and cnn+lstm/ ('3d_cnn') with the trainer's loop body, including
  clip_grad_norm_ (3dcnn/train_3D_Quadtree_cnn_model.py:111-125).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import PKG, ROOT, check_summary, pkg, rel_err

pytestmark = pytest.mark.gpu
# bf16: one rounding per Conv3d output now (27 taps accumulate in f32 inside ONE launch): measured 1.1e-3 .. 2.6e-3 on the
# fixtures (rounds 1-2 summed three launches through the bf16 map and needed 6e-2)
LOGIT_TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _oracle():
    sys.path.insert(0, ROOT)
    import oracle.quadtree_oracle as o
    return o


def _cos(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def _inputs(B, T, HW, salt):
    synth = pkg("synth")
    return (synth.synth_images(B * T, salt=salt, size=HW).view(B, T, 3, HW, HW),
            synth.synth_pose_features(B * T, salt=salt, realistic=True).view(B, T, 47), synth.synth_labels(B, 12, salt=salt))


def _to_tb(x_ncthw):
    """[B][C][T][H][W] -> time-major NHWC [T][B][H][W][C]"""
    return x_ncthw.permute(2, 0, 3, 4, 1).contiguous()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pt,T,H", [(1, 3, 10), (2, 5, 9), (2, 4, 8)])
def test_maxpool3d_forward_backward(dt, pt, T, H):
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, C, W = 2, 16, H + 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, T, H, W, generator=g).to(dt).float().requires_grad_(True)
    ref = F.max_pool3d(x, (pt, 2, 2), (pt, 2, 2))
    dref = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dref)
    To, Ho, Wo = T // pt, H // 2, W // 2
    xd = _to_tb(x.detach()).to(dev, dt)
    out = torch.empty(To, B, Ho, Wo, C, dtype=dt, device=dev)
    arg = torch.empty(To, B, Ho, Wo, C, dtype=torch.uint8, device=dev)
    L.check(lib.qt_pool3d_max(L.qt_dtype(dt), L.ptr(xd), L.ptr(out), L.ptr(arg), T, B, H, W, C, pt, L.stream_ptr()), "pool")
    assert torch.equal(out.float().cpu(), _to_tb(ref.detach()))
    dd = _to_tb(dref).to(dev, dt)
    dx = torch.full((T, B, H, W, C), 9.0, dtype=dt, device=dev)
    L.check(lib.qt_pool3d_max_bwd(L.qt_dtype(dt), L.ptr(dd), L.ptr(arg), L.ptr(dx), T, B, H, W, C, pt, L.stream_ptr()), "pool bwd")
    assert torch.equal(dx.float().cpu(), _to_tb(x.grad))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pt,T,H", [(1, 3, 10), (2, 5, 9), (2, 4, 8)])
def test_fused_bn_relu_maxpool3d_equals_separate_kernels(dt, pt, T, H):
    """BatchNorm3d -> ReLU -> MaxPool3d of one conv block (/root/reference/3D models/models.py:26-47, the nn.Sequential of
    Quadtree3DCNN's conv blocks) as one forward pass and one backward pass: the pooled map and argmax codes equal
    qt_bn_act + qt_pool3d_max bit for bit; (dy, dgamma, dbeta) equal qt_pool3d_max_bwd + qt_bn_bwd_reduce / finalize / apply
    (same per-element arithmetic; the channel sums differ only in summation order) and torch autograd in f32."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, C, W = 2, 16, H + 2
    q = L.qt_dtype(dt)
    st = L.stream_ptr()
    g = torch.Generator().manual_seed(11)
    y = torch.randn(B, C, T, H, W, generator=g).to(dt).float()
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.3
    mean = y.mean((0, 2, 3, 4))
    var = y.var((0, 2, 3, 4), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    To, Ho, Wo = T // pt, H // 2, W // 2
    M, cells = T * B * H * W, To * B * Ho * Wo
    yd = _to_tb(y).to(dev, dt)
    sc, sh, mu, isd, gm = (v.to(dev).contiguous() for v in (scale, shift, mean, invstd, gamma))

    # separate kernels
    a = torch.empty_like(yd)
    L.check(lib.qt_bn_act(q, L.ptr(yd), L.ptr(sc), L.ptr(sh), None, None, None, 1, L.ptr(a), ctypes.c_longlong(M), C, st), "act")
    out0 = torch.empty(To, B, Ho, Wo, C, dtype=dt, device=dev)
    arg0 = torch.empty(To, B, Ho, Wo, C, dtype=torch.uint8, device=dev)
    L.check(lib.qt_pool3d_max(q, L.ptr(a), L.ptr(out0), L.ptr(arg0), T, B, H, W, C, pt, st), "pool")
    # fused
    out1 = torch.empty_like(out0)
    arg1 = torch.empty_like(arg0)
    ymax = torch.empty_like(out0)
    L.check(lib.qt_pool3d_bn_relu_max(q, L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(out1), L.ptr(arg1), L.ptr(ymax), T, B, H, W, C, C,
                                      pt, st), "fused pool")
    torch.cuda.synchronize()
    assert torch.equal(out0, out1)
    assert torch.equal(arg0, arg1)
    # ymax is the raw conv output at the argmax position
    yv = yd[: To * pt, :, : Ho * 2, : Wo * 2].reshape(To, pt, B, Ho, 2, Wo, 2, C).permute(0, 2, 3, 5, 7, 1, 4, 6).reshape(To, B, Ho, Wo, C, pt * 4)
    assert torch.equal(ymax, torch.gather(yv, -1, arg1.long().unsqueeze(-1)).squeeze(-1))

    dd = torch.randn(To, B, Ho, Wo, C, generator=g).to(dev, dt)

    def finalize(part, rows):
        dgb = torch.empty(2, C, device=dev)
        coef = torch.empty(3, C, device=dev)
        L.check(lib.qt_bn_bwd_finalize(L.ptr(part), rows, C, ctypes.c_longlong(M), L.ptr(gm), L.ptr(isd), L.ptr(dgb[0]), L.ptr(dgb[1]),
                                       0, L.ptr(coef), st), "finalize")
        return dgb, coef

    # separate backward
    da = torch.empty_like(yd)
    L.check(lib.qt_pool3d_max_bwd(q, L.ptr(dd), L.ptr(arg0), L.ptr(da), T, B, H, W, C, pt, st), "pool bwd")
    rows0 = lib.qt_bn_bwd_partial_rows(ctypes.c_longlong(M), C)
    part0 = torch.empty(lib.qt_stats_capacity_rows(rows0), 2, C, device=dev)
    L.check(lib.qt_bn_bwd_reduce(q, L.ptr(da), L.ptr(a), L.ptr(yd), L.ptr(mu), L.ptr(isd), L.ptr(part0), ctypes.c_longlong(M), C, st),
            "reduce")
    dgb0, coef0 = finalize(part0, rows0)
    dy0 = torch.empty_like(yd)
    L.check(lib.qt_bn_bwd_apply(q, L.ptr(da), L.ptr(a), L.ptr(yd), L.ptr(mu), L.ptr(isd), L.ptr(coef0), L.ptr(dy0), None,
                                ctypes.c_longlong(M), C, st), "apply")
    # fused backward
    rows1 = lib.qt_bn_bwd_partial_rows(ctypes.c_longlong(cells), C)
    part1 = torch.empty(lib.qt_stats_capacity_rows(rows1), 2, C, device=dev)
    L.check(lib.qt_bn_bwd_reduce(q, L.ptr(dd), L.ptr(out1), L.ptr(ymax), L.ptr(mu), L.ptr(isd), L.ptr(part1), ctypes.c_longlong(cells),
                                 C, st), "reduce (pooled side)")
    dgb1, coef1 = finalize(part1, rows1)
    dy1 = torch.full_like(yd, 7.0)
    L.check(lib.qt_pool3d_bn_bwd_apply(q, L.ptr(dd), L.ptr(arg1), L.ptr(out1), L.ptr(yd), L.ptr(mu), L.ptr(isd), L.ptr(coef1),
                                       L.ptr(dy1), T, B, H, W, C, C, C, pt, st), "fused apply")
    torch.cuda.synchronize()
    assert torch.allclose(dgb0, dgb1, rtol=1e-5, atol=1e-5)
    tol = 2e-2 if dt == torch.bfloat16 else 1e-5
    assert torch.allclose(dy0.float(), dy1.float(), rtol=tol, atol=tol)
    # the same per-element arithmetic on the other path's coefficients is bit-identical
    dy2 = torch.empty_like(yd)
    L.check(lib.qt_pool3d_bn_bwd_apply(q, L.ptr(dd), L.ptr(arg1), L.ptr(out1), L.ptr(yd), L.ptr(mu), L.ptr(isd), L.ptr(coef0),
                                       L.ptr(dy2), T, B, H, W, C, C, C, pt, st), "fused apply")
    torch.cuda.synchronize()
    assert torch.equal(dy0, dy2)

    if dt == torch.bfloat16:   # the resident-grid form the large maps take (round 4): the same arithmetic, bit for bit
        lib.qt_set_pool3d_apply_light_min(ctypes.c_longlong(1))
        dy5 = torch.full_like(yd, 7.0)
        L.check(lib.qt_pool3d_bn_bwd_apply(q, L.ptr(dd), L.ptr(arg1), L.ptr(out1), L.ptr(yd), L.ptr(mu), L.ptr(isd), L.ptr(coef1),
                                           L.ptr(dy5), T, B, H, W, C, C, C, pt, st), "fused apply, resident grid")
        lib.qt_set_pool3d_apply_light_min(ctypes.c_longlong(0))
        torch.cuda.synchronize()
        assert torch.equal(dy1, dy5)

    # y rows narrower than the pooled rows (the first layer's 32 channels feeding 64-channel K rows): same values in the
    # first channels, zeros in the padding of pooled / argmax / y_at_max / dy
    Cy = C // 2
    yn = yd[..., :Cy].contiguous()
    out3, arg3, ymax3 = torch.full_like(out0, 5.0), torch.full_like(arg0, 5), torch.full_like(out0, 5.0)
    L.check(lib.qt_pool3d_bn_relu_max(q, L.ptr(yn), L.ptr(sc), L.ptr(sh), L.ptr(out3), L.ptr(arg3), L.ptr(ymax3), T, B, H, W, C, Cy,
                                      pt, st), "fused pool, narrow y")
    dy3 = torch.full_like(yd, 7.0)
    L.check(lib.qt_pool3d_bn_bwd_apply(q, L.ptr(dd), L.ptr(arg3), L.ptr(out3), L.ptr(yn), L.ptr(mu), L.ptr(isd), L.ptr(coef1),
                                       L.ptr(dy3), T, B, H, W, C, Cy, C, pt, st), "fused apply, narrow y")
    dy4 = torch.full_like(yn, 7.0)
    L.check(lib.qt_pool3d_bn_bwd_apply(q, L.ptr(dd), L.ptr(arg3), L.ptr(out3), L.ptr(yn), L.ptr(mu), L.ptr(isd), L.ptr(coef1),
                                       L.ptr(dy4), T, B, H, W, C, Cy, Cy, pt, st), "fused apply, narrow y and dy")
    torch.cuda.synchronize()
    for full, narrow in ((out1, out3), (arg1, arg3), (ymax, ymax3), (dy1, dy3)):
        assert torch.equal(full[..., :Cy], narrow[..., :Cy]) and (narrow[..., Cy:] == 0).all()
    assert torch.equal(dy4, dy1[..., :Cy])

    # torch autograd, f32, on the values the kernels saw
    yt = y.clone().requires_grad_(True)
    gt = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    ref = F.max_pool3d(torch.relu(F.batch_norm(yt, None, None, gt, bt, True, 0.0, 1e-5)), (pt, 2, 2), (pt, 2, 2))
    ref.backward(dd.float().cpu().permute(1, 4, 0, 2, 3))
    if dt == torch.float32:
        assert torch.allclose(out1.cpu(), _to_tb(ref.detach()), rtol=1e-5, atol=1e-5)
        assert torch.allclose(dgb1[0].cpu(), gt.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(dgb1[1].cpu(), bt.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(dy1.cpu(), _to_tb(yt.grad), rtol=1e-4, atol=1e-5)
    else:
        assert rel_err(dy1.float().cpu(), _to_tb(yt.grad)) < 2e-2
        assert rel_err(dgb1[0].cpu(), gt.grad) < 2e-2


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_bn_stats_pack_and_avgpool(dt):
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    lib.qt_bn_stats_rows.argtypes = [ctypes.c_longlong, ctypes.c_int]
    g = torch.Generator().manual_seed(6)
    for M, C in ((70001, 64), (513, 1024), (255, 8)):
        y = torch.randn(M, C, generator=g).to(dt)
        rows = lib.qt_bn_stats_rows(M, C)
        part = torch.zeros(rows, 2, C, device=dev)
        L.check(lib.qt_bn_stats(L.qt_dtype(dt), L.ptr(y.to(dev)), ctypes.c_longlong(M), C, L.ptr(part), L.stream_ptr()), "bn_stats")
        s = part.sum(0).cpu()
        yd = y.double()
        assert rel_err(s[0], yd.sum(0)) <= 1e-4 and rel_err(s[1], (yd * yd).sum(0)) <= 1e-5
    # clip packing: [B][T][3][H][W] -> [T][B][H][W][128] = the 27 taps x 3 channels of a pixel
    B, T, H, W = 2, 3, 6, 5
    clips = torch.randn(B, T, 3, H, W, generator=g)
    dst = torch.empty(T, B, H, W, 128, dtype=dt, device=dev)
    L.check(lib.qt_pack_clip27(L.qt_dtype(dt), L.ptr(clips.to(dev)), L.ptr(dst), B, T, H, W, L.stream_ptr()), "pack")
    xp = F.pad(clips.permute(0, 2, 1, 3, 4), (1, 1, 1, 1, 1, 1))            # [B][3][T+2][H+2][W+2]
    ref = torch.zeros(T, B, H, W, 128)
    for kt in range(3):
        for kh in range(3):
            for kw in range(3):
                k0 = ((kt * 3 + kh) * 3 + kw) * 3
                ref[..., k0:k0 + 3] = xp[:, :, kt:kt + T, kh:kh + H, kw:kw + W].permute(2, 0, 3, 4, 1)
    assert torch.equal(dst.float().cpu(), ref.to(dt).float())
    # AdaptiveAvgPool3d((1,1,1)) + flatten into columns of an f32 matrix, and its backward
    T, B, HW, C, ld, col0 = 2, 3, 12, 64, 100, 20
    x = torch.randn(T, B, HW, C, generator=g).to(dt)
    out = torch.zeros(B, ld, device=dev)
    L.check(lib.qt_avgpool_tb(L.qt_dtype(dt), L.ptr(x.to(dev)), L.ptr(out), T, B, HW, C, ld, col0, L.stream_ptr()), "avgpool_tb")
    assert rel_err(out[:, col0:col0 + C].cpu(), x.float().mean((0, 2))) <= 1e-5
    d = torch.randn(B, ld, generator=g)
    gx = torch.empty(T, B, HW, C, dtype=dt, device=dev)
    L.check(lib.qt_avgpool_tb_bwd(L.qt_dtype(dt), L.ptr(d.to(dev)), L.ptr(gx), T, B, HW, C, ld, col0, L.stream_ptr()), "avgpool_tb_bwd")
    want = (d[:, col0:col0 + C] / (T * HW)).view(1, B, 1, C).expand(T, B, HW, C)
    assert rel_err(gx.float().cpu(), want) <= (1e-6 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("cfg", [(2, 8, 16, 32), (3, 1, 8, 16), (1, 5, 12, 64), (2, 3, 4, 256)])
def test_first_conv3d_from_the_f32_clip(cfg):
    """conv3d_block1's nn.Conv3d(3, 32, 3x3x3, padding 1) (/root/reference/3dcnn/models.py:108) straight from the f32 clip
    [B][T][3][H][W] (qt_conv3d_first_fwd): against torch CPU fp32 conv3d on the bf16-rounded clip and filter.  T = 1 (both
    neighbour frames outside the clip), odd T, one 4-row slab per image, the widest row the kernel takes; BatchNorm3d partial
    sums of the f32 accumulator; the folded scale / shift / ReLU form; shapes it does not cover are refused."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, T, H, W = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(91)
    clip = torch.randn(B, T, 3, H, W, generator=g)
    w = (torch.randn(32, 3, 3, 3, 3, generator=g) * (2.0 / 81) ** 0.5)
    ref = F.conv3d(clip.to(dt).float().permute(0, 2, 1, 3, 4), w.to(dt).float(), None, 1, 1)    # [B][32][T][H][W]
    ref_tb = ref.permute(2, 0, 3, 4, 1).contiguous()                                              # [T][B][H][W][32]
    # the packed filter of qt_pack_conv3d_block(first = 1): [64][128], element ((kt*3 + kh)*3 + kw)*3 + c
    wp = torch.zeros(64, 128)
    wp[:32, :81] = w.permute(0, 2, 3, 4, 1).reshape(32, 81)
    wp = wp.to(dev, dt)
    cd = clip.to(dev)
    rows = lib.qt_conv3d_first_stats_rows(B, T, H, W)
    assert rows > 0
    y = torch.full((T, B, H, W, 32), float("nan"), dtype=dt, device=dev)
    part = torch.full((rows, 2, 64), float("nan"), device=dev)
    L.check(lib.qt_conv3d_first_fwd(L.qt_dtype(dt), L.ptr(cd), L.ptr(wp), L.ptr(y), None, None, 0, L.ptr(part), B, T, H, W,
                                    L.stream_ptr()), "qt_conv3d_first_fwd")
    torch.cuda.synchronize()
    assert rel_err(y.float().cpu(), ref_tb) <= 1e-2
    sums = part.sum(0).cpu()
    assert (sums[:, 32:] == 0).all()
    rd = ref_tb.double().reshape(-1, 32)
    assert rel_err(sums[0, :32], rd.sum(0)) <= 1e-4 and rel_err(sums[1, :32], (rd * rd).sum(0)) <= 1e-4
    # plain (no statistics) is the same map; scale / shift / ReLU is applied to the f32 accumulator
    y2 = torch.empty_like(y)
    L.check(lib.qt_conv3d_first_fwd(L.qt_dtype(dt), L.ptr(cd), L.ptr(wp), L.ptr(y2), None, None, 0, None, B, T, H, W, L.stream_ptr()),
            "plain")
    sc = (torch.rand(32, generator=g) + 0.5).to(dev)
    sh = (torch.randn(32, generator=g) * 0.3).to(dev)
    y3 = torch.empty_like(y)
    L.check(lib.qt_conv3d_first_fwd(L.qt_dtype(dt), L.ptr(cd), L.ptr(wp), L.ptr(y3), L.ptr(sc), L.ptr(sh), 1, None, B, T, H, W,
                                    L.stream_ptr()), "affine")
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert rel_err(y3.float().cpu(), torch.relu(ref_tb * sc.cpu() + sh.cpu())) <= 1e-2
    # the whole eval block in one launch: conv + folded BatchNorm3d + ReLU + MaxPool3d((1,2,2)) = the affine form, pooled
    want = torch.empty(T, B, H // 2, W // 2, 32, dtype=dt, device=dev)
    L.check(lib.qt_pool3d_max(L.qt_dtype(dt), L.ptr(y3), L.ptr(want), None, T, B, H, W, 32, 1, L.stream_ptr()), "pool")
    for pc in (64, 32):   # rows padded to 64 channels (zeros), and 32-channel rows (round 4)
        pooled = torch.full((T, B, H // 2, W // 2, pc), float("nan"), dtype=dt, device=dev)
        L.check(lib.qt_conv3d_first_fwd_pool(L.qt_dtype(dt), L.ptr(cd), L.ptr(wp), L.ptr(pooled), pc, L.ptr(sc), L.ptr(sh), B, T, H, W,
                                             L.stream_ptr()), "qt_conv3d_first_fwd_pool")
        torch.cuda.synchronize()
        assert torch.equal(pooled[..., :32], want) and (pooled[..., 32:] == 0).all()
    # weight gradient from the clip and a 32-channel-row gradient map, in nn.Conv3d's layout; deterministic
    lib.qt_conv3d_first_wgrad_workspace_bytes.restype = ctypes.c_size_t
    nws = int(lib.qt_conv3d_first_wgrad_workspace_bytes(B, T, H, W))
    assert (nws > 0) == (W % 32 == 0)
    if nws:
        dyt = torch.randn(T, B, H, W, 32, generator=g).to(dt)
        xr = clip.to(dt).float().permute(0, 2, 1, 3, 4).contiguous()
        wr = w.to(dt).float().requires_grad_(True)
        F.conv3d(xr, wr, None, 1, 1).backward(dyt.float().permute(1, 4, 0, 2, 3))
        dyd = dyt.to(dev)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        outs = []
        for _ in range(2):
            dw = torch.full((32, 3, 3, 3, 3), float("nan"), device=dev)
            L.check(lib.qt_conv3d_first_wgrad(L.qt_dtype(dt), L.ptr(cd), L.ptr(dyd), L.ptr(dw), L.ptr(ws), ctypes.c_size_t(nws), B, T, H,
                                              W, L.stream_ptr()), "qt_conv3d_first_wgrad")
            torch.cuda.synchronize()
            outs.append(dw.cpu())
        assert rel_err(outs[0], wr.grad) <= 2e-4        # products of bf16 values are exact in f32: only the summation order differs
        assert torch.equal(outs[0], outs[1])
        # round 4: the same weight gradient with dy formed inside the kernel (MaxPool3d((1,2,2)) + ReLU + BatchNorm3d backward of
        # the pooled gradient) against qt_pool3d_bn_bwd_apply + qt_conv3d_first_wgrad; pooled rows of 32 and of 64 channels
        mean = (torch.randn(32, generator=g) * 0.1).to(dev)
        invstd = (torch.rand(32, generator=g) + 0.5).to(dev)
        for cp in (32, 64):
            pooled = torch.empty(T, B, H // 2, W // 2, cp, dtype=dt, device=dev)
            arg = torch.empty(T, B, H // 2, W // 2, cp, dtype=torch.uint8, device=dev)
            L.check(lib.qt_pool3d_bn_relu_max(L.qt_dtype(dt), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(pooled), L.ptr(arg), None, T, B, H,
                                              W, cp, 32, 1, L.stream_ptr()), "qt_pool3d_bn_relu_max")
            dout = torch.randn(T, B, H // 2, W // 2, cp, generator=g).to(dt).to(dev)
            coef = torch.stack([torch.rand(cp, generator=g) + 0.5, torch.randn(cp, generator=g) * 0.05,
                                torch.randn(cp, generator=g) * 0.05]).to(dev)
            dy = torch.empty(T, B, H, W, 32, dtype=dt, device=dev)
            L.check(lib.qt_pool3d_bn_bwd_apply(L.qt_dtype(dt), L.ptr(dout), L.ptr(arg), L.ptr(pooled), L.ptr(y), L.ptr(mean),
                                               L.ptr(invstd), L.ptr(coef), L.ptr(dy), T, B, H, W, cp, 32, 32, 1, L.stream_ptr()),
                    "qt_pool3d_bn_bwd_apply")
            dw_ref = torch.full((32, 3, 3, 3, 3), float("nan"), device=dev)
            L.check(lib.qt_conv3d_first_wgrad(L.qt_dtype(dt), L.ptr(cd), L.ptr(dy), L.ptr(dw_ref), L.ptr(ws), ctypes.c_size_t(nws), B, T,
                                              H, W, L.stream_ptr()), "qt_conv3d_first_wgrad")
            dw_f = torch.full((32, 3, 3, 3, 3), float("nan"), device=dev)
            L.check(lib.qt_conv3d_first_wgrad_fused(L.qt_dtype(dt), L.ptr(cd), L.ptr(y), L.ptr(dout), L.ptr(arg), cp, L.ptr(mean),
                                                    L.ptr(invstd), L.ptr(sc), L.ptr(sh), L.ptr(coef), L.ptr(dw_f), L.ptr(ws),
                                                    ctypes.c_size_t(nws), B, T, H, W, L.stream_ptr()), "qt_conv3d_first_wgrad_fused")
            torch.cuda.synchronize()
            assert torch.isfinite(dw_f).all()
            # (dy = ka g + kb - y kc in the fused form against a (g - b - (y - mean) invstd c): single bf16 roundings differ)
            assert rel_err(dw_f.cpu(), dw_ref.cpu()) <= 2e-3, (cp, rel_err(dw_f.cpu(), dw_ref.cpu()))
        assert lib.qt_conv3d_first_wgrad_fused(L.qt_dtype(dt), L.ptr(cd), L.ptr(y), L.ptr(dout), L.ptr(arg), 48, L.ptr(mean),
                                               L.ptr(invstd), L.ptr(sc), L.ptr(sh), L.ptr(coef), L.ptr(dw_f), L.ptr(ws),
                                               ctypes.c_size_t(nws), B, T, H, W, L.stream_ptr()) == -1
    # not covered: f32, a width that is not a multiple of 16, a height that is not a multiple of 4
    assert lib.qt_conv3d_first_fwd(L.QT_F32, L.ptr(cd), L.ptr(wp), L.ptr(y), None, None, 0, None, B, T, H, W, L.stream_ptr()) == -3
    assert lib.qt_conv3d_first_stats_rows(B, T, H, W + 8) == 0 and lib.qt_conv3d_first_stats_rows(B, T, H + 2, W) == 0


@pytest.mark.parametrize("cfg", [(2, 8, 16, 32, 64), (3, 1, 8, 16, 64), (1, 5, 12, 64, 32), (2, 3, 4, 112, 64), (1, 2, 8, 128, 64)])
def test_second_conv3d_with_frame_slabs_in_lds(cfg):
    """conv3d_block2's nn.Conv3d(32, 64, 3x3x3, padding 1) (/root/reference/3dcnn/models.py:115) on the slab-resident kernel
    (qt_conv3d_c32_fwd): against torch CPU fp32 conv3d on the bf16-rounded operands.  T = 1, odd T, T = 2 (ring of three
    with two frames), one slab per image, the widest rows it takes, 32- and 64-channel input rows (the padding channels are
    never read: they hold garbage here); BatchNorm3d partial sums; the scale / shift / ReLU form; refused shapes."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, T, H, W, xc = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, 32, T, H, W, generator=g).to(dt).float()
    w = (torch.randn(64, 32, 3, 3, 3, generator=g) * (2.0 / (32 * 27)) ** 0.5).to(dt).float()
    ref = F.conv3d(x, w, None, 1, 1).permute(2, 0, 3, 4, 1).contiguous()            # [T][B][H][W][64]
    xd = torch.full((T, B, H, W, xc), 3.0, dtype=dt)                                  # garbage in the padding channels
    xd[..., :32] = x.permute(2, 0, 3, 4, 1).to(dt)
    xd = xd.to(dev)
    wp = torch.full((64, 27, 64), 3.0, dtype=dt)                                      # [O][tap][I padded]: garbage in the padding
    wp[:, :, :32] = w.permute(0, 2, 3, 4, 1).reshape(64, 27, 32).to(dt)
    wp = wp.to(dev)
    rows = lib.qt_conv3d_c32_stats_rows(B, T, H, W)
    assert rows > 0
    y = torch.full((T, B, H, W, 64), float("nan"), dtype=dt, device=dev)
    part = torch.full((rows, 2, 64), float("nan"), device=dev)
    L.check(lib.qt_conv3d_c32_fwd(L.qt_dtype(dt), L.ptr(xd), xc, L.ptr(wp), L.ptr(y), None, None, 0, L.ptr(part), B, T, H, W,
                                  L.stream_ptr()), "qt_conv3d_c32_fwd")
    torch.cuda.synchronize()
    assert rel_err(y.float().cpu(), ref) <= 1e-2
    sums = part.sum(0).cpu()
    rd = ref.double().reshape(-1, 64)
    assert rel_err(sums[0], rd.sum(0)) <= 1e-4 and rel_err(sums[1], (rd * rd).sum(0)) <= 1e-4
    y2 = torch.empty_like(y)
    L.check(lib.qt_conv3d_c32_fwd(L.qt_dtype(dt), L.ptr(xd), xc, L.ptr(wp), L.ptr(y2), None, None, 0, None, B, T, H, W,
                                  L.stream_ptr()), "plain")
    sc = (torch.rand(64, generator=g) + 0.5).to(dev)
    sh = (torch.randn(64, generator=g) * 0.3).to(dev)
    y3 = torch.empty_like(y)
    L.check(lib.qt_conv3d_c32_fwd(L.qt_dtype(dt), L.ptr(xd), xc, L.ptr(wp), L.ptr(y3), L.ptr(sc), L.ptr(sh), 1, None, B, T, H, W,
                                  L.stream_ptr()), "affine")
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert rel_err(y3.float().cpu(), torch.relu(ref * sc.cpu() + sh.cpu())) <= 1e-2
    # the data gradient: the same walk over dy with the flipped filter [I][27][O], two passes joined through an f32 scratch
    dyt = torch.randn(T, B, H, W, 64, generator=g).to(dt)
    xr = x.clone().requires_grad_(True)
    F.conv3d(xr, w, None, 1, 1).backward(dyt.float().permute(1, 4, 0, 2, 3))
    wdp = torch.full((64, 27, 64), 3.0, dtype=dt)                                     # rows 32..63 (padding input channels): garbage
    wdp[:32] = w.permute(1, 2, 3, 4, 0).reshape(32, 27, 64).to(dt)
    wdp = wdp.to(dev)
    lib.qt_conv3d_c32_dgrad_scratch_bytes.restype = ctypes.c_size_t
    nscr = int(lib.qt_conv3d_c32_dgrad_scratch_bytes(B, T, H, W))
    assert nscr == T * B * H * W * 32 * 4
    scr = torch.empty(nscr, dtype=torch.uint8, device=dev)
    dyd = dyt.to(dev)
    for dxc in (64, 32):   # rows padded to 64 channels (32..63 written as zeros), and 32-channel rows (round 4)
        dx = torch.full((T, B, H, W, dxc), float("nan"), dtype=dt, device=dev)
        L.check(lib.qt_conv3d_c32_dgrad(L.qt_dtype(dt), L.ptr(dyd), L.ptr(wdp), L.ptr(dx), dxc, L.ptr(scr), ctypes.c_size_t(nscr), B, T,
                                        H, W, L.stream_ptr()), "qt_conv3d_c32_dgrad")
        torch.cuda.synchronize()
        assert rel_err(dx[..., :32].float().cpu(), xr.grad.permute(2, 0, 3, 4, 1)) <= 1e-2
        assert (dx[..., 32:] == 0).all()
    assert lib.qt_conv3d_c32_dgrad(L.qt_dtype(dt), L.ptr(dyd), L.ptr(wdp), L.ptr(dx), 48, L.ptr(scr), ctypes.c_size_t(nscr), B, T, H, W,
                                   L.stream_ptr()) == -1
    # the weight gradient: contraction over positions, nn.Conv3d's layout, deterministic
    wr = w.clone().requires_grad_(True)
    F.conv3d(x, wr, None, 1, 1).backward(dyt.float().permute(1, 4, 0, 2, 3))
    lib.qt_conv3d_c32_wgrad_workspace_bytes.restype = ctypes.c_size_t
    nws = int(lib.qt_conv3d_c32_wgrad_workspace_bytes(B, T, H, W))
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        dwt = torch.full((64, 32, 3, 3, 3), float("nan"), device=dev)
        L.check(lib.qt_conv3d_c32_wgrad(L.qt_dtype(dt), L.ptr(xd), xc, L.ptr(dyd), L.ptr(dwt), L.ptr(ws), ctypes.c_size_t(nws), B, T, H, W,
                                        L.stream_ptr()), "qt_conv3d_c32_wgrad")
        torch.cuda.synchronize()
        outs.append(dwt.cpu())
    assert rel_err(outs[0], wr.grad) <= 2e-4       # products of bf16 values are exact in f32: only the summation order differs
    assert torch.equal(outs[0], outs[1])
    assert lib.qt_conv3d_c32_fwd(L.QT_F32, L.ptr(xd), xc, L.ptr(wp), L.ptr(y), None, None, 0, None, B, T, H, W, L.stream_ptr()) == -3
    assert lib.qt_conv3d_c32_stats_rows(B, T, H, W + 8) == 0 and lib.qt_conv3d_c32_stats_rows(B, T, H + 1, W) == 0
    assert lib.qt_conv3d_c32_stats_rows(B, T, H, 144) == 0


CASES = [("q3_t8", 2, 8, 112, "quadtree_3d_fusion", 31), ("q3_t5", 2, 5, 64, "quadtree_3d_fusion", 31),
         ("q3_img_t8", 2, 8, 64, "quadtree_3d_image_only", 31), ("ji_t4", 2, 4, 64, None, 32)]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 8, 64, 64, 28), (3, 5, 64, 128, 14), (1, 1, 128, 64, 16), (2, 4, 128, 256, 28)])
def test_conv3d_as_one_27_tap_launch(dt, cfg):
    """nn.Conv3d(3x3x3, padding 1) as ONE implicit GEMM over 27 taps (qt_conv_desc.kt = 3) on time-major clips: forward with
    bias and BatchNorm3d statistics, and the data gradient, against torch CPU fp32 conv3d on the same (pre-rounded)
    operands.  T = 1 (every neighbour frame masked), odd T, several clips per frame."""
    dev = _dev()
    L = pkg("_lib")
    B, T, Cin, Cout, H = cfg
    g = torch.Generator().manual_seed(71)
    x = torch.randn(B, Cin, T, H, H, generator=g).to(dt).float()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5).to(dt).float()
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv3d(x, w, bias, 1, 1)                                           # [B][Cout][T][H][H]
    xd = x.permute(2, 0, 3, 4, 1).contiguous().to(dev, dt)                     # [T][B][H][W][C]
    wf = w.permute(0, 2, 3, 4, 1).contiguous().to(dev, dt)                     # [O][kt][kh][kw][I]
    wd = w.permute(1, 2, 3, 4, 0).contiguous().to(dev, dt)                     # [I][kt][kh][kw][O]
    tol = 2e-5 if dt == torch.float32 else 1.5e-2

    def desc(mode, kin, kout):
        d = L.ConvDesc()
        d.dtype = L.qt_dtype(dt); d.mode = mode; d.batch = T * B
        d.in_h = d.in_w = d.out_h = d.out_w = H
        d.k_per_tap, d.n_out = kin, kout
        d.kh = d.kw = 3; d.stride = 1; d.pad = 1
        d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * kin, H * kin, kin
        d.kt, d.frames = 3, T
        return d
    d = desc(L.QT_CONV_FWD, Cin, Cout)
    rows = L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
    y = torch.full((T * B * H * H, Cout), float("nan"), dtype=dt, device=dev)
    st = torch.zeros(rows, 2, Cout, device=dev)
    io = L.ConvIO(L.ptr(xd), L.ptr(wf), L.ptr(y), None, L.ptr(bias.to(dev)), None, None, L.ptr(st))
    L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm kt=3")
    torch.cuda.synchronize()
    got = y.float().cpu().view(T, B, H, H, Cout).permute(1, 4, 0, 2, 3)
    assert rel_err(got, ref) <= tol
    # statistics of the raw accumulator (before the bias): sums of conv3d(x, w)
    raw = ref - bias.view(1, -1, 1, 1, 1)
    s = st.sum(0).cpu()
    assert rel_err(s[0], raw.sum((0, 2, 3, 4))) <= 1e-3 + tol
    assert rel_err(s[1], (raw * raw).sum((0, 2, 3, 4))) <= 1e-3 + tol
    # data gradient
    dy = torch.randn(B, Cout, T, H, H, generator=g).to(dt).float()
    dxr = torch.nn.grad.conv3d_input((B, Cin, T, H, H), w, dy, 1, 1)
    dyd = dy.permute(2, 0, 3, 4, 1).contiguous().to(dev, dt)
    dd = desc(L.QT_CONV_DGRAD, Cout, Cin)
    dx = torch.full((T * B * H * H, Cin), float("nan"), dtype=dt, device=dev)
    io = L.ConvIO(L.ptr(dyd), L.ptr(wd), L.ptr(dx), None, None, None, None, None)
    L.check(L.lib().qt_conv2d_igemm(ctypes.byref(dd), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm kt=3 dgrad")
    torch.cuda.synchronize()
    assert rel_err(dx.float().cpu().view(T, B, H, H, Cin).permute(1, 4, 0, 2, 3), dxr) <= tol
    # frame taps are refused where they are not implemented
    bad = desc(L.QT_CONV_FWD, Cin, Cout)
    bad.frames = T + 1 if (T * B) % (T + 1) else 0
    assert L.lib().qt_conv2d_igemm(ctypes.byref(bad), ctypes.byref(io), L.stream_ptr()) != 0


def _build(mode, T, dt, dropout=0.0):
    P, synth = pkg(), pkg("synth")
    m = P.Ji3DCNN(12, sequence_length=T, dropout_rate=dropout, compute_dtype=dt) if mode is None else \
        P.Quadtree3DCNN(12, sequence_length=T, mode=mode, dropout_rate=dropout, compute_dtype=dt)
    m.load_state_dict(synth.synth_state_dict(m))
    return m


def _is_conv_bias_before_bn(name):
    return name.endswith(".0.bias") and (name.startswith("conv3d_") or name.startswith("visual_stream."))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag,B,T,HW,mode,salt", CASES)
def test_clip_models_match_reference_golden(dt, tag, B, T, HW, mode, salt, golden_clip3d):
    dev = _dev()
    g = golden_clip3d
    x, f, y = _inputs(B, T, HW, salt)
    m = _build(mode, T, dt).to(dev).eval()
    with torch.no_grad():
        logits = m(x.to(dev), f.to(dev))
    print(f"{tag} {dt}: eval logits rel err {rel_err(logits.cpu(), g[f'{tag}/eval/logits']):.2e}")
    assert rel_err(logits.cpu(), g[f"{tag}/eval/logits"]) <= LOGIT_TOL[dt]
    m.train()
    out = m(x.to(dev), f.to(dev))
    loss = F.cross_entropy(out, y.to(dev))
    loss.backward()
    assert rel_err(out.detach().cpu(), g[f"{tag}/train/logits"]) <= LOGIT_TOL[dt]
    assert abs(loss.item() - float(g[f"{tag}/train/loss"])) <= (1e-3 if dt == torch.float32 else 2e-2) * max(1.0, abs(float(g[f"{tag}/train/loss"])))
    params = dict(m.named_parameters())
    worst_cos = 1.0
    for name in [str(n) for n in g[f"{tag}/train/grad_names"]]:
        grad = params[name].grad
        assert grad is not None and torch.isfinite(grad).all(), name
        gold_s = g[f"{tag}/train/grad/{name}/sample"]
        if _is_conv_bias_before_bn(name):   # exactly zero in exact arithmetic (the batch mean removes the bias): bounded
            wkey = f"{tag}/train/grad/{name[:-4]}weight"
            wmean = float(g[wkey + "/abssum"]) / float(np.prod(g[wkey + "/shape"]))   # mean |weight gradient| of the layer
            assert float(grad.abs().max()) <= (2.0 if dt == torch.float32 else 100.0) * wmean + 1e-5, name
            continue
        fl = grad.detach().double().flatten().cpu()
        stride = max(1, fl.numel() // 256)
        smp = fl[::stride][:256].numpy()
        head = name.split(".")[0] in ("classifier", "numerical_projection", "numerical_lstm")
        if float(np.abs(gold_s).max()) == 0.0:   # (frame taps that never see a source frame, T = 1: exactly zero)
            assert float(np.abs(smp).max()) == 0.0, name
            continue
        if dt == torch.float32:
            # (conv gradients pass ReLU / max-pool decisions: a flipped decision moves everything below it)
            assert _cos(smp, gold_s) >= (0.9999 if head else 0.999), (name, _cos(smp, gold_s))
            assert float(np.abs(smp - gold_s).max()) <= (2e-3 if head else 6e-2) * max(float(np.abs(gold_s).max()), 1e-30), name
        else:
            cosv = _cos(smp, gold_s)
            worst_cos = min(worst_cos, cosv) if not head else worst_cos
            assert cosv >= (0.98 if head else 0.9), (name, cosv)
    if dt == torch.bfloat16:
        print(f"{tag} bf16: smallest cosine of a conv gradient against the reference {worst_cos:.4f}")
    bufs = dict(m.named_buffers())
    for k in g.files:
        if k.startswith(f"{tag}/train/buf/") and k.endswith("/shape"):
            n = k[len(f"{tag}/train/buf/"):-len("/shape")]
            check_summary(bufs[n].cpu(), g, f"{tag}/train/buf/{n}", 1e-4 if dt == torch.float32 else 3e-2)


def test_quadtree3d_config4_size_matches_oracle():
    """BASELINE config 4 at its own size: clips of T = 8 frames of 224x224 (f32 build against the CPU oracle on the box;
    bf16 build: stated bound) -- eval logits and the dropout-free train step's loss."""
    dev = _dev()
    o = _oracle()
    B, T, HW = 2, 8, 224
    x, f, y = _inputs(B, T, HW, 77)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m = _build("quadtree_3d_fusion", T, torch.float32)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = o.quadtree3d_forward(sd0, x, f)
    sd = o.clip_params(sd0)
    ref_loss = F.cross_entropy(o.quadtree3d_forward(sd, x, f, train=True, dropout_p=0.0), y)
    for dt in (torch.float32, torch.bfloat16):
        mm = _build("quadtree_3d_fusion", T, dt).to(dev).eval()
        with torch.no_grad():
            got = mm(x.to(dev), f.to(dev)).cpu()
        assert rel_err(got, ref) <= LOGIT_TOL[dt], (dt, rel_err(got, ref))
        mm.train()
        loss = F.cross_entropy(mm(x.to(dev), f.to(dev)), y.to(dev))
        loss.backward()
        assert abs(loss.item() - ref_loss.item()) <= (1e-3 if dt == torch.float32 else 2e-2) * max(1.0, abs(ref_loss.item()))
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mm.parameters())
        del mm
        torch.cuda.empty_cache()


def test_block1_pooled_map_in_32_channel_rows(monkeypatch):
    """Round 4: where conv3d_block2 runs on the slab kernels, conv3d_block1's pooled map, its argmax / raw-value companions and
    the gradient coming back are 32-channel rows (video3d._ConvBlock._pooled_width) instead of rows padded to 64: the same
    forward bits, and the same gradients up to the order of the BatchNorm-backward partial sums."""
    dev = _dev()
    v3d = pkg("video3d")
    B, T, HW = 2, 4, 64
    x, f, y = (t.to(dev) for t in _inputs(B, T, HW, 5))
    res = []
    for narrow, fused in ((True, True), (False, True), (True, False), (False, False)):
        # (fused: conv3d_block1's dy formed inside its weight-gradient kernel, qt_conv3d_first_wgrad_fused)
        monkeypatch.setattr(v3d, "POOLED32", narrow)
        monkeypatch.setattr(v3d, "FIRST_WGRAD_FUSED", fused)
        m = _build("quadtree_3d_fusion", T, torch.bfloat16).to(dev).train()
        widths = []
        orig = v3d._Ops.pool_bn

        def spy(self, dt, y_, stats, out, *a, **k):
            widths.append(out.shape[1])
            return orig(self, dt, y_, stats, out, *a, **k)
        monkeypatch.setattr(v3d._Ops, "pool_bn", spy)
        out = m(x, f)
        F.cross_entropy(out, y).backward()
        monkeypatch.setattr(v3d._Ops, "pool_bn", orig)
        assert widths[0] == (32 if narrow else 64), widths
        res.append((out.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for n, g32 in res[0][1].items():
            if _is_conv_bias_before_bn(n):
                continue
            assert rel_err(g32.cpu(), other[1][n].cpu()) <= 2e-3, n


def test_dropout_train_mode_and_eval_backward_and_errors():
    dev = _dev()
    P = pkg()
    QtError = pkg("_lib").QtError
    B, T, HW = 2, 4, 32
    x, f, y = _inputs(B, T, HW, 5)
    m = _build("quadtree_3d_fusion", T, torch.bfloat16, dropout=0.6).to(dev).train()
    out = m(x.to(dev), f.to(dev))
    F.cross_entropy(out, y.to(dev)).backward()
    assert torch.isfinite(out).all() and all(torch.isfinite(p.grad).all() for p in m.parameters())
    with pytest.raises(QtError):
        m(x, f)                                            # CPU tensors: no fallback
    with pytest.raises(ValueError):
        m(x.to(dev)[:, :, :2], f.to(dev))                  # not 3 channels
    with pytest.raises(ValueError):
        P.Quadtree3DCNN(12, mode="bogus")
    # the numerical sequence may arrive on the CPU (3dcnn/train_3D_Quadtree_cnn_model.py:113 moves only the clips)
    m.eval()
    with torch.no_grad():
        a = m(x.to(dev), f)
        b = m(x.to(dev), f.to(dev))
    assert torch.equal(a, b)


def test_packed_operands_follow_the_parameters(monkeypatch):
    """The conv blocks re-pack their filter / BatchNorm vectors only when one of them changed (video3d._ConvBlock.pack): an
    in-place edit through torch, a step of the package's FusedAdam (raw pointers: optim.raw_update_count) and a step of
    torch's fused Adam (bumps nothing: invalidated by the backward that produced its gradients) must all reach the next
    forward -- logits equal to those of a forward that re-packs unconditionally."""
    dev = _dev()
    P = pkg()
    v3d = pkg("video3d")
    B, T, HW = 2, 4, 32
    x, f, y = (t.to(dev) for t in _inputs(B, T, HW, 9))
    m = _build("quadtree_3d_fusion", T, torch.bfloat16).to(dev)

    def eval_logits(cache):
        monkeypatch.setattr(v3d, "PACK_CACHE", cache)
        m.eval()
        with torch.no_grad():
            return m(x, f).clone()

    a0 = eval_logits(True)
    assert torch.equal(a0, eval_logits(True)) and torch.equal(a0, eval_logits(False))
    w = m.conv3d_block2[0].weight
    with torch.no_grad():
        w.mul_(0.5)
    a1 = eval_logits(True)
    assert not torch.equal(a1, a0) and torch.equal(a1, eval_logits(False))
    with torch.no_grad():
        w.mul_(2.0)                                        # (exact: back to the first weights)
    assert torch.equal(eval_logits(True), a0)
    # a write through `.data` bumps no version counter the cache can see: model.invalidate_packed() is the documented way
    w.data.mul_(0.5)
    assert torch.equal(eval_logits(True), a0)              # (stale on purpose: that is the limitation)
    m.invalidate_packed()
    assert torch.equal(eval_logits(True), a1)
    w.data.mul_(2.0)
    m.invalidate_packed()
    assert torch.equal(eval_logits(True), a0)
    for make in (lambda ps: P.FusedAdam(ps, lr=1e-2), lambda ps: torch.optim.Adam(ps, lr=1e-2, fused=True)):
        opt = make(list(m.parameters()))
        before = eval_logits(True)
        monkeypatch.setattr(v3d, "PACK_CACHE", True)
        m.train()
        opt.zero_grad(set_to_none=True)
        F.cross_entropy(m(x, f), y).backward()
        opt.step()
        after = eval_logits(True)
        assert not torch.equal(after, before) and torch.equal(after, eval_logits(False))


class _sibling_models:
    def __init__(self, sub):
        self.dir = os.path.join(ROOT, PKG, sub)

    def __enter__(self):
        import importlib
        sys.modules.pop("models", None)
        sys.path.insert(0, self.dir)
        return importlib.import_module("models")

    def __exit__(self, *exc):
        sys.path.remove(self.dir)
        sys.modules.pop("models", None)


def test_dropin_3dcnn_trainer_loop(monkeypatch, tmp_path, capsys):
    """3dcnn/train_3D_Quadtree_cnn_model.py:80-86,111-125: get_model(mode=..., sequence_length=...), zero_grad, forward,
    CrossEntropyLoss, backward, clip_grad_norm_(model.parameters(), 1.0), Adam step; then cnn+lstm's '3d_cnn'."""
    dev = _dev()
    o = _oracle()
    synth = pkg("synth")
    monkeypatch.setenv("QTCNN_DTYPE", "f32")
    B, T, HW = 2, 5, 64
    x, f, y = _inputs(B, T, HW, 31)
    with _sibling_models("threed_cnn") as models:
        model = models.get_model(num_classes=12, device=dev, mode="quadtree_3d_fusion", sequence_length=T)
        assert "(Mode: quadtree_3d_fusion)" in capsys.readouterr().out
        for mod in model.modules():
            if type(mod).__name__ == "Dropout":
                mod.p = 0.0
        model.dropout_rate = 0.0
        model.numerical_lstm.dropout = 0.0
        model.load_state_dict({k: v.to(dev) for k, v in synth.synth_state_dict(model).items()})
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        criterion = torch.nn.CrossEntropyLoss()
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
        model.train()
        losses = []
        for _ in range(2):
            optimizer.zero_grad()
            outputs = model(x.to(dev), f)          # (the trainer leaves the numerical sequence where the loader put it)
            loss = criterion(outputs, y.to(dev))
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            optimizer.step()
            losses.append(loss.item())
        sd = o.clip_params(sd0)
        opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-4, weight_decay=1e-4)
        ref = []
        for _ in range(2):
            opt.zero_grad()
            l = F.cross_entropy(o.quadtree3d_forward(sd, x, f, train=True, dropout_p=0.0), y)
            l.backward()
            torch.nn.utils.clip_grad_norm_([v for v in sd.values() if v.requires_grad], 1.0)
            opt.step()
            ref.append(l.item())
        assert abs(losses[0] - ref[0]) <= 1e-3 * max(1.0, abs(ref[0]))
        assert abs(losses[1] - ref[1]) <= 3e-2 * max(1.0, abs(ref[1]))
        path = os.path.join(tmp_path, "m.pth")
        torch.save(model.state_dict(), path)
        fresh = models.get_model(num_classes=12, device=dev, mode="quadtree_3d_fusion", sequence_length=T, print_num_params=False)
        fresh.load_state_dict(torch.load(path, map_location=dev))
        model.eval(); fresh.eval()
        with torch.no_grad():
            assert torch.equal(model(x.to(dev), f), fresh(x.to(dev), f))
        for bad in ("resnet_3d_video_only", "hybrid_quadtree_3d_fusion"):
            with pytest.raises(NotImplementedError):
                models.get_model(num_classes=12, device=dev, mode=bad)
    with _sibling_models("cnn_lstm") as models:
        ji = models.get_model("3d_cnn", 12, dev, seq_len=4)
        assert type(ji).__name__ == "Ji3DCNN"
        xj, fj, yj = _inputs(2, 4, 64, 32)
        ji.eval()
        with torch.no_grad():
            assert ji(xj.to(dev), fj.to(dev)).shape == (2, 12)
