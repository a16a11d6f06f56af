"""3-D clip models on the HIP kernels (BASELINE config 4).

Drop-in counterparts of
  Quadtree3DCNN  /root/reference/3dcnn/models.py:96-214   (mode quadtree_3d_fusion / quadtree_3d_image_only)
  Ji3DCNN        /root/reference/cnn+lstm/models.py:93-142
with the reference's constructor arguments, attribute tree, state_dict keys and forward signature
(image_sequence [B,T,3,H,W], numerical_sequence [B,T,47]) -> logits [B,C].

How a Conv3d runs here.  Clip activations are TIME-MAJOR NHWC, [T][B][H][W][C], in the compute dtype (bf16, or f32
with QTCNN_DTYPE=f32).  A 3x3x3 convolution with padding 1 is ONE implicit GEMM over 27 taps (qt_conv_desc.kt = 3,
csrc/conv_igemm.hip: tap (kt, kh, kw) of an output pixel of frame t reads frame t + kt - 1, masked per row where that
frame does not exist), forward and data gradient alike: f32 accumulation over all taps, the epilogue adds the bias and
emits the BatchNorm3d statistics.  The weight gradient is one launch per frame tap over the contiguous range of frames
the tap connects (the tile-resident bf16 kernel with fixed-order partial sums where it covers the shape).  The first
layer (3 input channels) is packed to one 128-wide K row per pixel (27 taps x 3 channels, qt_pack_clip27) and runs as a
1x1 convolution; its 32 output channels are padded to 64 with zero filters so that the next layer's K rows are whole
128-byte chunks.  MaxPool3d / AdaptiveAvgPool3d come from csrc/video3d.hip, the LSTM recurrences from csrc/lstm.hip, the
thin dense products from csrc/gemm_small.hip.  The whole forward / backward is ONE autograd node; PyTorch only owns the
buffers.  There is no torch / CPU fallback.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib
from . import modules as M
from . import optim as _optim
from ._lib import QtError
from .engine import default_compute_dtype

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
_c = ctypes


class _GemmDesc(ctypes.Structure):   # qt_gemm_small_desc
    _fields_ = [("M", _c.c_int), ("N", _c.c_int), ("K", _c.c_int),
                ("a_dtype", _c.c_int), ("b_dtype", _c.c_int), ("c_dtype", _c.c_int),
                ("a_row_stride", _c.c_longlong), ("a_k_stride", _c.c_longlong),
                ("b_row_stride", _c.c_longlong), ("b_k_stride", _c.c_longlong),
                ("c_row_stride", _c.c_longlong), ("relu", _c.c_int), ("accumulate", _c.c_int)]


class _BnEvalItem(ctypes.Structure):   # qt_bn_eval_item
    _fields_ = [("gamma", _c.c_void_p), ("beta", _c.c_void_p), ("running_mean", _c.c_void_p), ("running_var", _c.c_void_p),
                ("scale", _c.c_void_p), ("shift", _c.c_void_p), ("C", _c.c_int), ("mean", _c.c_void_p),
                ("invstd", _c.c_void_p)]


def _ptr(t, byte_offset=0):
    return None if t is None else _c.c_void_p(t.data_ptr() + byte_offset)


class _Ops:
    """ctypes view of the C ABI used by the clip models; every call checks its status."""

    def __init__(self):
        self.L = _lib.lib()
        L = self.L
        L.qt_stats_capacity_rows.restype = _c.c_int
        L.qt_conv2d_wgrad_workspace_bytes.restype = _c.c_size_t
        self._wgrad_ws = None
        self.timed = None   # list while bench.py profiles: (start event, end event, algorithmic flops, bytes, mode) per conv launch
        L.qt_bn_stats_rows.argtypes = [_c.c_longlong, _c.c_int]
        L.qt_bn_bwd_partial_rows.argtypes = [_c.c_longlong, _c.c_int]

    def check(self, rc, what):
        _lib.check(rc, what)

    # ---- convolutions -------------------------------------------------------------------------------------
    @staticmethod
    def conv_desc(dt, mode, images, h, w, k_per_tap, n_out, k, pad):
        d = _lib.ConvDesc()
        d.dtype = _lib.qt_dtype(dt)
        d.mode = mode
        d.batch = images
        d.in_h, d.in_w, d.out_h, d.out_w = h, w, h, w
        d.k_per_tap, d.n_out = k_per_tap, n_out
        d.kh = d.kw = k
        d.stride, d.pad = 1, pad
        d.src_img_stride, d.src_row_stride, d.src_pix_stride = h * w * k_per_tap, w * k_per_tap, k_per_tap
        return d

    def igemm(self, d, src, w, dst, scale=None, shift=None, residual=None, relu=0, stats=None, flops=0.0, nbytes=0.0):
        d.relu = relu
        io = _lib.ConvIO(src, w, dst, _ptr(scale), _ptr(shift), residual, None, _ptr(stats))
        ev = None
        if self.timed is not None:   # bench.py's roofline: HIP events on the launch stream (= torch's current stream here)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.check(self.L.qt_conv2d_igemm(_c.byref(d), _c.byref(io), _lib.stream_ptr()), "qt_conv2d_igemm")
        if ev is not None:
            ev[1].record()
            self.timed.append((ev[0], ev[1], float(flops), float(nbytes), int(d.mode)))

    def wgrad(self, d, dy, x, dw):
        """dw [n_out][taps][k_per_tap] f32 (zeroed by the caller).  Where the tile-resident kernel covers the shape (bf16,
        3x3 / stride 1) it gets its partial-filter workspace: fixed-order sums instead of float atomics."""
        nbytes = self.L.qt_conv2d_wgrad_workspace_bytes(_c.byref(d))
        if nbytes > 0:
            if self._wgrad_ws is None or self._wgrad_ws.numel() < nbytes or self._wgrad_ws.device != dw.device:
                self._wgrad_ws = torch.empty(nbytes, dtype=torch.uint8, device=dw.device)
            self.check(self.L.qt_conv2d_wgrad_ws(_c.byref(d), dy, x, _ptr(dw), _ptr(self._wgrad_ws), _c.c_size_t(nbytes),
                                                 _lib.stream_ptr()), "qt_conv2d_wgrad_ws")
        else:
            self.check(self.L.qt_conv2d_wgrad(_c.byref(d), dy, x, _ptr(dw), _lib.stream_ptr()), "qt_conv2d_wgrad")

    def bn_finalize(self, part, prow, Mrows, C, gamma, beta, rmean, rvar, nbt, dev):
        out = torch.empty(4, C, dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
        self.check(self.L.qt_bn_finalize(_ptr(part), prow, C, _c.c_longlong(Mrows), _ptr(gamma), _ptr(beta), _ptr(rmean),
                                         _ptr(rvar), _ptr(nbt), _c.c_float(BN_MOMENTUM), _c.c_float(BN_EPS),
                                         _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _lib.stream_ptr()),
                   "qt_bn_finalize")
        return out

    def pack_weight(self, dt, w_oihw, w_fwd, w_dgrad, O, I, k):
        self.check(self.L.qt_pack_conv_weight(_lib.qt_dtype(dt), _ptr(w_oihw), _ptr(w_fwd), _ptr(w_dgrad), O, I, k, k,
                                              _lib.stream_ptr()), "qt_pack_conv_weight")

    def unpack_wgrad(self, dw, grad_oihw, O, I, k):
        self.check(self.L.qt_unpack_conv_wgrad(_ptr(dw), _ptr(grad_oihw), O, I, k, k, 0, _lib.stream_ptr()),
                   "qt_unpack_conv_wgrad")

    # ---- BatchNorm / activation / pooling -------------------------------------------------------------------
    def bn_train(self, dt, y, Mrows, C, gamma, beta, rmean, rvar, nbt, dev):
        rows = self.L.qt_bn_stats_rows(_c.c_longlong(Mrows), C)
        part = torch.empty(self.L.qt_stats_capacity_rows(rows), 2, C, dtype=torch.float32, device=dev)
        self.check(self.L.qt_bn_stats(_lib.qt_dtype(dt), _ptr(y), _c.c_longlong(Mrows), C, _ptr(part), _lib.stream_ptr()),
                   "qt_bn_stats")
        out = torch.empty(4, C, dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
        self.check(self.L.qt_bn_finalize(_ptr(part), rows, C, _c.c_longlong(Mrows), _ptr(gamma), _ptr(beta), _ptr(rmean),
                                         _ptr(rvar), _ptr(nbt), _c.c_float(BN_MOMENTUM), _c.c_float(BN_EPS),
                                         _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _lib.stream_ptr()),
                   "qt_bn_finalize")
        return out

    def bn_eval(self, gamma, beta, rmean, rvar, C, dev):
        out = torch.empty(4, C, dtype=torch.float32, device=dev)   # running mean, 1/sqrt(running var + eps), scale, shift
        item = _BnEvalItem(_ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), _ptr(out[2]), _ptr(out[3]), C, _ptr(out[0]),
                           _ptr(out[1]))
        self.check(self.L.qt_bn_eval_affine_batched(_c.byref(item), 1, _c.c_float(BN_EPS), _lib.stream_ptr()),
                   "qt_bn_eval_affine_batched")
        return out

    def bn_act(self, dt, y, stats, out, Mrows, C):
        self.check(self.L.qt_bn_act(_lib.qt_dtype(dt), _ptr(y), _ptr(stats[2]), _ptr(stats[3]), None, None, None, 1, _ptr(out),
                                    _c.c_longlong(Mrows), C, _lib.stream_ptr()), "qt_bn_act")

    def bn_backward(self, dt, g, act, y, stats, gamma, Mrows, C, dev, batch_stats):
        """g = d/d(relu(bn(y))) -> (dy, dgamma, dbeta); `act` is the ReLU output (its mask)."""
        rows = self.L.qt_bn_bwd_partial_rows(_c.c_longlong(Mrows), C)
        part = torch.empty(self.L.qt_stats_capacity_rows(rows), 2, C, dtype=torch.float32, device=dev)
        q = _lib.qt_dtype(dt)
        self.check(self.L.qt_bn_bwd_reduce(q, _ptr(g), _ptr(act), _ptr(y), _ptr(stats[0]), _ptr(stats[1]), _ptr(part),
                                           _c.c_longlong(Mrows), C, _lib.stream_ptr()), "qt_bn_bwd_reduce")
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        self.check(self.L.qt_bn_bwd_finalize(_ptr(part), rows, C, _c.c_longlong(Mrows if batch_stats else 0), _ptr(gamma),
                                             _ptr(stats[1]), _ptr(dgb[0]), _ptr(dgb[1]), 0, _ptr(coef), _lib.stream_ptr()),
                   "qt_bn_bwd_finalize")
        dy = torch.empty_like(y)
        self.check(self.L.qt_bn_bwd_apply(q, _ptr(g), _ptr(act), _ptr(y), _ptr(stats[0]), _ptr(stats[1]), _ptr(coef), _ptr(dy),
                                          None, _c.c_longlong(Mrows), C, _lib.stream_ptr()), "qt_bn_bwd_apply")
        return dy, dgb[0], dgb[1]

    def pack_clip(self, dt, clip, B, T, H, W):
        x = torch.empty(T * B * H * W, 128, dtype=dt, device=clip.device)
        self.check(self.L.qt_pack_clip27(_lib.qt_dtype(dt), _ptr(clip), _ptr(x), B, T, H, W, _lib.stream_ptr()), "qt_pack_clip27")
        return x

    def conv3d_first(self, dt, clip, wf, y, part, B, T, H, W, flops, nbytes, pooled=None):
        ev = None
        if self.timed is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if pooled is not None:   # y is the pooled map; pooled = the folded BatchNorm3d's (.., .., scale, shift)
            self.check(self.L.qt_conv3d_first_fwd_pool(_lib.qt_dtype(dt), _ptr(clip), _ptr(wf), _ptr(y), y.shape[1], _ptr(pooled[2]),
                                                       _ptr(pooled[3]), B, T, H, W, _lib.stream_ptr()), "qt_conv3d_first_fwd_pool")
        else:
            self.check(self.L.qt_conv3d_first_fwd(_lib.qt_dtype(dt), _ptr(clip), _ptr(wf), _ptr(y), None, None, 0, _ptr(part), B, T, H, W,
                                                  _lib.stream_ptr()), "qt_conv3d_first_fwd")
        if ev is not None:
            ev[1].record()
            self.timed.append((ev[0], ev[1], float(flops), float(nbytes), int(_lib.QT_CONV_FWD)))

    def conv3d_c32(self, dt, x, xc, wf, y, B, T, H, W, scale=None, shift=None, relu=0, stats=None, flops=0.0, nbytes=0.0):
        ev = None
        if self.timed is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.check(self.L.qt_conv3d_c32_fwd(_lib.qt_dtype(dt), _ptr(x), xc, _ptr(wf), _ptr(y), _ptr(scale), _ptr(shift), relu, _ptr(stats),
                                            B, T, H, W, _lib.stream_ptr()), "qt_conv3d_c32_fwd")
        if ev is not None:
            ev[1].record()
            self.timed.append((ev[0], ev[1], float(flops), float(nbytes), int(_lib.QT_CONV_FWD)))

    def conv3d_c32_dgrad(self, dt, dy, wd, dx, dxc, scr, nscr, B, T, H, W, flops=0.0, nbytes=0.0):
        ev = None
        if self.timed is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.check(self.L.qt_conv3d_c32_dgrad(_lib.qt_dtype(dt), _ptr(dy), _ptr(wd), _ptr(dx), dxc, _ptr(scr), _c.c_size_t(nscr), B, T,
                                              H, W, _lib.stream_ptr()), "qt_conv3d_c32_dgrad")
        if ev is not None:
            ev[1].record()
            self.timed.append((ev[0], ev[1], float(flops), float(nbytes), int(_lib.QT_CONV_DGRAD)))

    def pool(self, dt, x, out, arg, T, B, H, W, C, pt):
        self.check(self.L.qt_pool3d_max(_lib.qt_dtype(dt), _ptr(x), _ptr(out), _ptr(arg), T, B, H, W, C, pt,
                                        _lib.stream_ptr()), "qt_pool3d_max")

    def pool_bn(self, dt, y, stats, out, arg, ymax, T, B, H, W, C, pt, cy=None):
        """BatchNorm3d (scale / shift) + ReLU + MaxPool3d in one pass over the raw conv output (csrc/video3d.hip); y rows are
        cy <= C channels wide"""
        self.check(self.L.qt_pool3d_bn_relu_max(_lib.qt_dtype(dt), _ptr(y), _ptr(stats[2]), _ptr(stats[3]), _ptr(out), _ptr(arg),
                                                _ptr(ymax), T, B, H, W, C, cy or C, pt, _lib.stream_ptr()), "qt_pool3d_bn_relu_max")

    def pool_bn_backward(self, dt, dout, arg, pooled, ymax, y, stats, gamma, T, B, H, W, C, pt, dev, batch_stats, cy=None, cd=None,
                         apply=True):
        """d/d(pooled) -> (dy, dgamma, dbeta): the BatchNorm sums from the pooled side (every cell sends its gradient to one
        position), then max-pool backward + ReLU mask + BatchNorm backward in one pass; no full-size gradient map in between.
        apply = False: no dy, the coefficients [3][C] instead (the consumer forms dy itself: qt_conv3d_first_wgrad_fused)"""
        cells = (T // pt) * B * (H // 2) * (W // 2)
        Mrows = T * B * H * W
        rows = self.L.qt_bn_bwd_partial_rows(_c.c_longlong(cells), C)
        part = torch.empty(self.L.qt_stats_capacity_rows(rows), 2, C, dtype=torch.float32, device=dev)
        q = _lib.qt_dtype(dt)
        self.check(self.L.qt_bn_bwd_reduce(q, _ptr(dout), _ptr(pooled), _ptr(ymax), _ptr(stats[0]), _ptr(stats[1]), _ptr(part),
                                           _c.c_longlong(cells), C, _lib.stream_ptr()), "qt_bn_bwd_reduce")
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        self.check(self.L.qt_bn_bwd_finalize(_ptr(part), rows, C, _c.c_longlong(Mrows if batch_stats else 0), _ptr(gamma),
                                             _ptr(stats[1]), _ptr(dgb[0]), _ptr(dgb[1]), 0, _ptr(coef), _lib.stream_ptr()),
                   "qt_bn_bwd_finalize")
        if not apply:
            return coef, dgb[0], dgb[1]
        cy, cd = cy or C, cd or cy or C
        dy = torch.empty(Mrows, cd, dtype=dt, device=dev)
        self.check(self.L.qt_pool3d_bn_bwd_apply(q, _ptr(dout), _ptr(arg), _ptr(pooled), _ptr(y), _ptr(stats[0]), _ptr(stats[1]),
                                                 _ptr(coef), _ptr(dy), T, B, H, W, C, cy, cd, pt, _lib.stream_ptr()),
                   "qt_pool3d_bn_bwd_apply")
        return dy, dgb[0], dgb[1]

    def pool_bwd(self, dt, dout, arg, dx, T, B, H, W, C, pt):
        self.check(self.L.qt_pool3d_max_bwd(_lib.qt_dtype(dt), _ptr(dout), _ptr(arg), _ptr(dx), T, B, H, W, C, pt,
                                            _lib.stream_ptr()), "qt_pool3d_max_bwd")

    def col_sum(self, dt, x, rows, cols, ld, out):
        self.check(self.L.qt_col_sum(_lib.qt_dtype(dt), _ptr(x), _c.c_longlong(rows), cols, ld, _ptr(out), 0, _lib.stream_ptr()),
                   "qt_col_sum")

    # ---- thin dense products (f32) --------------------------------------------------------------------------
    def gemm(self, Mr, N, K, A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, c_rs, bias=None, relu=0, a_off=0, c_off=0):
        d = _GemmDesc(Mr, N, K, _lib.QT_F32, _lib.QT_F32, _lib.QT_F32, a_rs, a_ks, b_rs, b_ks, c_rs, relu, 0)
        self.check(self.L.qt_gemm_small(_c.byref(d), _ptr(A, a_off * 4), _ptr(Bm), _ptr(bias), _ptr(Cm, c_off * 4),
                                        _lib.stream_ptr()), "qt_gemm_small")

    def linear(self, x, x_ld, x_off, W, b, rows, out, out_ld, out_off, relu):
        """out[r, out_off:out_off+N] = relu?(x[r, x_off:x_off+K] W^T + b)  (nn.Linear; W [N,K])"""
        N, K = W.shape
        self.gemm(rows, N, K, x, x_ld, 1, W, K, 1, out, out_ld, b, relu, x_off, out_off)

    def dropout(self, x, rows, cols, ld, off, seed, p):
        self.check(self.L.qt_dropout(_lib.QT_F32, _ptr(x, off * 4), _c.c_longlong(rows), cols, ld, _c.c_ulonglong(seed),
                                     _c.c_float(p), _lib.stream_ptr()), "qt_dropout")


# QTCNN_POOL3D_FUSED (default 1): conv blocks with a pool run BatchNorm3d + ReLU + MaxPool3d as one pass forward and one
# pass backward (csrc/video3d.hip); 0: qt_bn_act + qt_pool3d_max / qt_pool3d_max_bwd + qt_bn_bwd_reduce + qt_bn_bwd_apply
FUSED_POOL = os.environ.get("QTCNN_POOL3D_FUSED", "1") != "0"

# QTCNN_LSTM_SIDE_STREAM (default 1): Quadtree3DCNN's LSTM branch (a dozen latency-bound launches of 32 workgroups, 0.3 ms
# forward and 0.25 ms backward in a row) runs on a second stream beside the conv blocks, which it does not depend on
LSTM_SIDE = os.environ.get("QTCNN_LSTM_SIDE_STREAM", "1") != "0"
# QTCNN_WGRAD_SIDE_STREAM (default 1): the conv blocks' weight gradients run on a stream of their own beside the data-gradient /
# BatchNorm / pooling chain of the backward pass (0: in line, as until round 3)
WGRAD_SIDE = os.environ.get("QTCNN_WGRAD_SIDE_STREAM", "1") != "0"
# QTCNN_PACK_CACHE (default 1): the clip models re-pack a conv block's filter / BatchNorm vectors only when one of them changed
# (writes through `p.data` are not seen: model.invalidate_packed(), see there)
PACK_CACHE = os.environ.get("QTCNN_PACK_CACHE", "1") != "0"
# QTCNN_CONV3D_SLAB (default 1): conv3d_block2's forward on the slab-resident kernel (csrc/conv3d_slab.hip); 0: 27-tap implicit GEMM
SLAB_C32 = os.environ.get("QTCNN_CONV3D_SLAB", "1") != "0"
# QTCNN_FIRST_WGRAD_FUSED (default 1): conv3d_block1's backward forms d(loss)/dy inside the weight-gradient kernel
# (qt_conv3d_first_wgrad_fused) instead of writing it with qt_pool3d_bn_bwd_apply and reading it back
FIRST_WGRAD_FUSED = os.environ.get("QTCNN_FIRST_WGRAD_FUSED", "1") != "0"
# QTCNN_POOLED32 (default 1): conv3d_block1's pooled map (and its argmax / raw-value companions, and the gradient block 2 sends
# back) in 32-channel rows where block 2 runs on the slab kernels, which read 32 channels; 0: rows padded to 64 channels (round 3)
POOLED32 = os.environ.get("QTCNN_POOLED32", "1") != "0"
_side_streams = {}


class _Side:
    """fork / join of a per-device second stream around a branch; tensors made there and used on the main stream are
    recorded on it (the caching allocator's pools are per stream)"""

    def __init__(self, dev, slot=0):
        self.main = torch.cuda.current_stream(dev)
        key = ((dev.index if dev.index is not None else torch.cuda.current_device()), slot)
        if key not in _side_streams:
            _side_streams[key] = torch.cuda.Stream(device=dev)
        self.side = _side_streams[key]

    def fork(self):
        self.side.wait_stream(self.main)
        return torch.cuda.stream(self.side)

    def join(self, *tensors):
        self.main.wait_stream(self.side)
        for t in tensors:
            if t is not None:
                t.record_stream(self.main)


_ops = None


def ops():
    global _ops
    if _ops is None:
        _ops = _Ops()
    return _ops


def _cpad(c):
    """channel count the conv kernels see: K rows are whole 128-byte chunks (64 bf16 / 32 f32 -> 64 covers both)"""
    return max(64, (c + 63) // 64 * 64)


class _ConvBlock:
    """Conv3d(3x3x3, pad 1, bias) + BatchNorm3d + ReLU (+ MaxPool3d (pt,2,2)) on [T][B][H][W][C].
    ONE forward and ONE data-gradient launch per Conv3d: the implicit GEMM walks all 27 taps (qt_conv_desc.kt = 3, f32
    accumulation, frames past the clip's ends masked per output row), its epilogue adds the bias and emits the BatchNorm3d
    statistics of the finished value.  (Rounds 1-2 ran three launches per convolution that accumulated through the bf16
    output map: two extra read + write passes per layer and two intermediate roundings.)"""

    def __init__(self, conv, bn, pool_t, first):
        self.conv, self.bn, self.pool_t, self.first = conv, bn, pool_t, first
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.cin_p = 128 if first else _cpad(self.cin)
        self.cout_p = _cpad(self.cout)
        self.narrow_out = False   # set by the model: the next block is the 32 -> 64 one (reads 32 channels on the slab kernels)
        self._buf_key = None

    # -- operand packing: ONE launch per forward (weights may have changed: fused optimizers do not bump _version) --
    def pack(self, dt, need_dgrad, epoch=0):
        o, dev = ops(), self.conv.weight.device
        key = (dt, dev, bool(need_dgrad))
        # nothing to do when neither the parameters nor the running statistics changed since the last pack (an eval loop:
        # five launches per forward, 8 % of Quadtree3DCNN's eval forward).  torch bumps a tensor's version on every in-place
        # op; FusedAdam's raw-pointer updates are counted by optim.raw_update_count(); torch's fused optimizers bump nothing, so
        # `epoch` (the model's count of backward passes) invalidates the copies whenever gradients were produced -- the rule of
        # engine.Engine.pack_weights
        tens = (self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        ver = tuple((t._version, t.data_ptr()) for t in tens) + (_optim.raw_update_count(), epoch)
        if PACK_CACHE and self._buf_key == key and getattr(self, "_packed_ver", None) == ver:
            return
        self._packed_ver = ver
        if self._buf_key != key:
            nf = self.cout_p * 128 if self.first else self.cout_p * 27 * self.cin_p
            self.wf = torch.empty(nf, dtype=dt, device=dev)
            self.wd = torch.empty(nf, dtype=dt, device=dev) if (need_dgrad and not self.first) else None
            self.vec = torch.empty(5, self.cout_p, dtype=torch.float32, device=dev)
            self._buf_key = key
        vals = (self.conv.bias.data_ptr(), self.bn.weight.data_ptr(), self.bn.bias.data_ptr(),
                self.bn.running_mean.data_ptr(), self.bn.running_var.data_ptr())
        if getattr(self, "_ptr_vals", None) != vals:   # (device array of the five vector pointers: rebuilt only when a tensor moved)
            self._ptrs = torch.tensor(vals, dtype=torch.int64).to(dev)
            self._ptr_vals = vals
        ptrs = self._ptrs
        o.check(o.L.qt_pack_conv3d_block(_lib.qt_dtype(dt), _ptr(self.conv.weight.detach()), _ptr(self.wf), _ptr(self.wd),
                                         self.cout, self.cin, self.cout_p, self.cin_p, 1 if self.first else 0, _ptr(ptrs),
                                         _ptr(self.vec), _lib.stream_ptr()), "qt_pack_conv3d_block")
        self.bias_p, self.gamma_p, self.beta_p, self.rmean_p, self.rvar_p = (self.vec[i] for i in range(5))

    def _desc(self, dt, mode, T, B, H, W):
        if self.first:
            return o_desc(dt, mode, T * B, H, W, 128, self.cout_p, 1, 0)
        kin, kout = (self.cin_p, self.cout_p) if mode == _lib.QT_CONV_FWD else (self.cout_p, self.cin_p)
        d = o_desc(dt, mode, T * B, H, W, kin, kout, 3, 1)
        d.kt, d.frames = 3, T
        return d

    def _vec_grad(self, v):
        """a per-channel gradient vector without its padding channels; no copy launch when there is no padding (every block
        but the first: three launches per block and step)"""
        return v if v.shape[0] == self.cout and v.is_contiguous() else v[:self.cout].clone()

    def _pooled_width(self, dt, B, T, H, W):
        """channels per row of this block's pooled map [T][B][H][W][.]: 32 (no padding) when the next block's forward, data
        gradient and weight gradient all run on the slab kernels at that size, else the padded width every other kernel reads"""
        if not (POOLED32 and self.narrow_out and SLAB_C32 and dt == torch.bfloat16 and self.cout == 32):
            return self.cout_p
        L = ops().L
        L.qt_conv3d_c32_dgrad_scratch_bytes.restype = _c.c_size_t
        ok = (L.qt_conv3d_c32_stats_rows(B, T, H, W) > 0 and int(L.qt_conv3d_c32_dgrad_scratch_bytes(B, T, H, W)) > 0
              and self._slab_wgrad_bytes(B, T, H, W) > 0)
        return 32 if ok else self.cout_p

    @staticmethod
    def _slab_wgrad_bytes(B, T, H, W):
        L = ops().L
        L.qt_conv3d_c32_wgrad_workspace_bytes.restype = _c.c_size_t
        return int(L.qt_conv3d_c32_wgrad_workspace_bytes(B, T, H, W))

    def _raw_rows(self, dt, x, T, B, H, W):
        """the first layer from the f32 clip itself (csrc/conv3d_first.hip): partial-sum rows, 0 = take the packed form"""
        if not (self.first and FUSED_POOL and self.pool_t and dt == torch.bfloat16 and self.cin == 3 and self.cout == 32):
            return 0
        if x.data_ptr() % 16:
            return 0
        return ops().L.qt_conv3d_first_stats_rows(B, T, H, W) if os.environ.get("QTCNN_CONV3D_FIRST", "1") != "0" else 0

    def _forward_raw(self, dt, clip, T, B, H, W, training, keep, prow):
        """conv3d_block1 without the packed K rows: conv from the f32 clip (y: 32-channel rows, bias-free), then BatchNorm3d +
        ReLU + MaxPool3d in one pass into the 64-channel rows the next layer reads"""
        o, dev = ops(), clip.device
        rows = T * B * H * W
        y = torch.empty(rows, 32, dtype=dt, device=dev) if (training or keep or self.pool_t != 1) else None
        fl = 2.0 * rows * 27 * self.cin * self.cout
        nb = 4.0 * rows * 3 + 2.0 * rows * 32
        if training:
            part = torch.empty(o.L.qt_stats_capacity_rows(prow), 2, self.cout_p, dtype=torch.float32, device=dev)
            o.conv3d_first(dt, clip, self.wf, y, part, B, T, H, W, fl, nb)
            stats = o.bn_finalize(part, prow, rows, self.cout_p, self.gamma_p, self.beta_p, self.rmean_p, self.rvar_p,
                                  self.bn.num_batches_tracked, dev)
            self.rmean_p.add_(self.bias_p, alpha=BN_MOMENTUM)
            self.bn.running_mean.copy_(self.rmean_p[:self.cout])
            self.bn.running_var.copy_(self.rvar_p[:self.cout])
        else:
            stats = o.bn_eval(self.gamma_p, self.beta_p, self.rmean_p, self.rvar_p, self.cout_p, dev)
            # y is the bias-free accumulator: BatchNorm3d(y + bias) = y * scale + (shift + scale * bias), xhat = (y - (mean - bias)) invstd
            stats[3].addcmul_(stats[2], self.bias_p)
            stats[0].sub_(self.bias_p)
            if not keep and self.pool_t == 1:   # eval without backward: the whole block in one launch, y never exists
                out = torch.empty(T * B * (H // 2) * (W // 2), self._pooled_width(dt, B, T, H // 2, W // 2), dtype=dt, device=dev)
                o.conv3d_first(dt, clip, self.wf, out, None, B, T, H, W, fl, 4.0 * rows * 3 + 2.0 * out.numel(), pooled=stats)
                return out, (T, H // 2, W // 2), None
            o.conv3d_first(dt, clip, self.wf, y, None, B, T, H, W, fl, nb)
        To, Ho, Wo = T // self.pool_t, H // 2, W // 2
        cp = self._pooled_width(dt, B, To, Ho, Wo)
        out = torch.empty(To * B * Ho * Wo, cp, dtype=dt, device=dev)
        arg = torch.empty(To * B * Ho * Wo, cp, dtype=torch.uint8, device=dev) if keep else None
        ymax = torch.empty_like(out) if keep else None
        o.pool_bn(dt, y, stats, out, arg, ymax, T, B, H, W, cp, self.pool_t, cy=32)
        saved = (clip, y, None, arg, stats, (T, B, H, W), training, out, ymax) if keep else None
        return out, (To, Ho, Wo), saved

    def forward(self, dt, x, T, B, H, W, training, keep):
        o, dev = ops(), x.device
        rows = T * B * H * W
        if self.first:   # x is the f32 clip [B][T][3][H][W]
            prow = self._raw_rows(dt, x, T, B, H, W)
            if prow > 0:
                return self._forward_raw(dt, x, T, B, H, W, training, keep, prow)
            x = o.pack_clip(dt, x, B, T, H, W)
        d = self._desc(dt, _lib.QT_CONV_FWD, T, B, H, W)
        y = torch.empty(rows, self.cout_p, dtype=dt, device=dev)
        fused_eval = not training and not keep   # eval without backward: BatchNorm3d + ReLU in the conv epilogue, no second pass
        esz = 2 if dt == torch.bfloat16 else 4
        fl = 2.0 * rows * 27 * self.cin * self.cout            # algorithmic: the Conv3d as the reference computes it
        nb = esz * (rows * (self.cin_p + self.cout_p) + 27.0 * self.cin_p * self.cout_p)
        tk = dict(flops=fl, nbytes=nb)
        # conv3d_block2 (32 -> 64 channels) with its frame slabs resident in LDS (csrc/conv3d_slab.hip) where the shape fits:
        # slab = its partial-sum rows, 0 = the 27-tap implicit GEMM
        slab = 0
        if (SLAB_C32 and dt == torch.bfloat16 and self.cin == 32 and self.cout == 64 and self.cin_p == 64 and self.cout_p == 64
                and x.data_ptr() % 16 == 0):
            slab = o.L.qt_conv3d_c32_stats_rows(B, T, H, W)
        if slab:
            tk = dict(flops=fl, nbytes=esz * (rows * (self.cin + self.cout_p) + 27.0 * self.cin * self.cout_p))

            def conv(**kw):
                o.conv3d_c32(dt, x, x.shape[1], self.wf, y, B, T, H, W, **kw, **tk)
        else:
            if x.shape[1] != self.cin_p:
                raise QtError(f"conv block: input rows of {x.shape[1]} channels, the implicit GEMM reads {self.cin_p}")

            def conv(**kw):
                o.igemm(d, _ptr(x), _ptr(self.wf), _ptr(y), **kw, **tk)
        if training:
            prow = slab or o.L.qt_conv2d_stats_rows(_c.byref(d))
            part = torch.empty(o.L.qt_stats_capacity_rows(prow), 2, self.cout_p, dtype=torch.float32, device=dev)
            # Under batch statistics BatchNorm3d(conv + bias) = BatchNorm3d(conv): a per-channel constant moves the mean with it.
            # y holds the bias-free accumulator and the epilogue's statistics are of exactly that value; the bias only enters
            # the running mean the reference tracks (mean of conv + bias), added below.
            conv(stats=part)
            stats = o.bn_finalize(part, prow, rows, self.cout_p, self.gamma_p, self.beta_p, self.rmean_p, self.rvar_p,
                                  self.bn.num_batches_tracked, dev)
            self.rmean_p.add_(self.bias_p, alpha=BN_MOMENTUM)
            self.bn.running_mean.copy_(self.rmean_p[:self.cout])
            self.bn.running_var.copy_(self.rvar_p[:self.cout])
        else:
            stats = o.bn_eval(self.gamma_p, self.beta_p, self.rmean_p, self.rvar_p, self.cout_p, dev)
            if fused_eval:   # relu(scale * (conv + bias) + shift)
                shift = torch.addcmul(stats[3], stats[2], self.bias_p)
                conv(scale=stats[2], shift=shift, relu=1)
            elif slab:   # (its epilogue has no shift-only form: conv * 1 + bias)
                conv(scale=torch.ones_like(self.bias_p), shift=self.bias_p)
            else:
                conv(shift=self.bias_p)
        ymax = None
        if self.pool_t and not fused_eval and FUSED_POOL:
            # BatchNorm3d + ReLU + MaxPool3d in one pass: relu(bn(y)) is read by nothing but the pool (the next block takes the
            # pooled map, the backward's ReLU mask is `pooled > 0` at the argmax), so it is not materialised
            a = None
            To, Ho, Wo = T // self.pool_t, H // 2, W // 2
            out = torch.empty(To * B * Ho * Wo, self.cout_p, dtype=dt, device=dev)
            arg = torch.empty(To * B * Ho * Wo, self.cout_p, dtype=torch.uint8, device=dev) if keep else None
            ymax = torch.empty_like(out) if keep else None
            o.pool_bn(dt, y, stats, out, arg, ymax, T, B, H, W, self.cout_p, self.pool_t)
        else:
            if fused_eval:
                a = y
            else:
                a = torch.empty_like(y)
                o.bn_act(dt, y, stats, a, rows, self.cout_p)
            arg, out, To, Ho, Wo = None, a, T, H, W
            if self.pool_t:
                To, Ho, Wo = T // self.pool_t, H // 2, W // 2
                out = torch.empty(To * B * Ho * Wo, self.cout_p, dtype=dt, device=dev)
                arg = torch.empty(To * B * Ho * Wo, self.cout_p, dtype=torch.uint8, device=dev) if keep else None
                o.pool(dt, a, out, arg, T, B, H, W, self.cout_p, self.pool_t)
        saved = (x, y, a, arg, stats, (T, B, H, W), training, out, ymax) if keep else None
        return out, (To, Ho, Wo), saved

    @staticmethod
    def _ranges(T, kt):
        """(dst frame begin, src frame begin, frames) of out[t] += conv(in[t + kt - 1])"""
        lo = max(0, 1 - kt)
        hi = min(T, T + 1 - kt)
        return lo, lo + kt - 1, max(0, hi - lo)

    def backward(self, dt, dout, saved, wside=None):
        """dout: d/d(block output) -> (dx or None, dW, db, dgamma, dbeta).  wside: a _Side whose stream takes the weight
        gradient (the caller joins it before dW is used)"""
        o = ops()
        x, y, a, arg, stats, (T, B, H, W), training, pooled, ymax = saved
        dev = x.device
        esz = 2 if dt == torch.bfloat16 else 4
        rows = T * B * H * W
        raw = self.first and x.dtype == torch.float32 and x.dim() == 5   # saved by _forward_raw: the clip itself, y 32 wide
        raw_ws, coef = 0, None
        if raw:
            o.L.qt_conv3d_first_wgrad_workspace_bytes.restype = _c.c_size_t
            raw_ws = int(o.L.qt_conv3d_first_wgrad_workspace_bytes(B, T, H, W))
            # (the pooled side's row width: 32 where block 2 runs on the slab kernels, else padded -- _pooled_width)
            fused_dy = bool(FIRST_WGRAD_FUSED and raw_ws and self.pool_t == 1 and dt == torch.bfloat16 and dout.data_ptr() % 16 == 0)
            dy, dgamma, dbeta = o.pool_bn_backward(dt, dout, arg, pooled, ymax, y, stats, self.gamma_p, T, B, H, W, pooled.shape[1],
                                                   self.pool_t, dev, training, cy=32, cd=32 if raw_ws else self.cout_p,
                                                   apply=not fused_dy)
            coef = dy if fused_dy else None   # (apply = False returns the coefficients in dy's place)
            if not raw_ws:   # (a width the raw weight-gradient kernel does not take: the packed rows after all)
                x = o.pack_clip(dt, x, B, T, H, W)
        elif ymax is not None:
            dy, dgamma, dbeta = o.pool_bn_backward(dt, dout, arg, pooled, ymax, y, stats, self.gamma_p, T, B, H, W, self.cout_p,
                                                   self.pool_t, dev, training)
        else:
            if self.pool_t:
                da = torch.empty_like(a)
                o.pool_bwd(dt, dout, arg, da, T, B, H, W, self.cout_p, self.pool_t)
            else:
                da = dout
            dy, dgamma, dbeta = o.bn_backward(dt, da, a, y, stats, self.gamma_p, rows, self.cout_p, dev, training)
        # conv bias gradient = column sums of dy.  No pass over dy is needed (it was 0.5 ms of the step): with
        # dy = ca (g - cb - xhat cc), ca = gamma invstd, the sum over positions is ca (sum g - M cb - cc sum xhat);
        # under batch statistics cb = sum g / M and sum xhat = 0: the gradient of a bias in front of a train-mode
        # BatchNorm is zero (the reference's autograd returns rounding noise there); under running statistics
        # cb = cc = 0 and it is gamma invstd sum g = gamma invstd dbeta.
        nv = dbeta.shape[0]   # (cout_p, or cout where the pooled side has no padding channels)
        if training:
            db = torch.zeros(nv, dtype=torch.float32, device=dev)
        else:
            db = self.gamma_p.detach().float()[:nv] * stats[1][:nv] * dbeta
        slab = (SLAB_C32 and dt == torch.bfloat16 and self.cin == 32 and self.cout == 64 and self.cin_p == 64 and self.cout_p == 64
                and dy.data_ptr() % 16 == 0)

        def weight_gradient():
            """dW from (x, dy): the kernels below only read what the chain above has produced"""
            dW = torch.empty_like(self.conv.weight)
            if raw_ws:
                ws = torch.empty(raw_ws, dtype=torch.uint8, device=dev)
                if coef is not None:   # dy = the pool / ReLU / BatchNorm3d backward of dout, formed inside the kernel
                    o.check(o.L.qt_conv3d_first_wgrad_fused(_lib.qt_dtype(dt), _ptr(x), _ptr(y), _ptr(dout), _ptr(arg), dout.shape[1],
                                                            _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(coef),
                                                            _ptr(dW), _ptr(ws), _c.c_size_t(raw_ws), B, T, H, W, _lib.stream_ptr()),
                            "qt_conv3d_first_wgrad_fused")
                    return dW
                o.check(o.L.qt_conv3d_first_wgrad(_lib.qt_dtype(dt), _ptr(x), _ptr(dy), _ptr(dW), _ptr(ws), _c.c_size_t(raw_ws), B, T,
                                                  H, W, _lib.stream_ptr()), "qt_conv3d_first_wgrad")
                return dW
            if self.first:
                d = self._desc(dt, _lib.QT_CONV_FWD, T, B, H, W)
                dw = torch.zeros(self.cout_p, 128, dtype=torch.float32, device=dev)
                o.wgrad(d, _ptr(dy), _ptr(x), dw)
            elif slab and x.data_ptr() % 16 == 0 and self._slab_wgrad_bytes(B, T, H, W):
                # conv3d_block2: weight gradient on the slab-resident kernel (csrc/conv3d_slab.hip)
                nws = self._slab_wgrad_bytes(B, T, H, W)
                ws = torch.empty(nws, dtype=torch.uint8, device=dev)
                o.check(o.L.qt_conv3d_c32_wgrad(_lib.qt_dtype(dt), _ptr(x), x.shape[1], _ptr(dy), _ptr(dW), _ptr(ws), _c.c_size_t(nws), B,
                                                T, H, W, _lib.stream_ptr()), "qt_conv3d_c32_wgrad")
                return dW
            else:
                if x.shape[1] != self.cin_p:
                    raise QtError(f"conv block: input rows of {x.shape[1]} channels, the tile weight gradient reads {self.cin_p}")
                # one launch per frame tap over the contiguous range of frames the tap connects (the contraction runs over
                # pixels: f32 sums, nothing accumulates through an activation map); with a workspace the bf16 build takes the
                # tile-resident kernel and its fixed-order partial sums: deterministic, no float atomics
                frame_in, frame_out = B * H * W * self.cin_p * esz, B * H * W * self.cout_p * esz
                dw = torch.zeros(3, self.cout_p, 9, self.cin_p, dtype=torch.float32, device=dev)
                for kt in range(3):
                    dlo, slo, cnt = self._ranges(T, kt)   # forward: out[dlo + i] read in[slo + i]
                    if cnt == 0:
                        continue
                    d2 = o_desc(dt, _lib.QT_CONV_FWD, cnt * B, H, W, self.cin_p, self.cout_p, 3, 1)
                    o.wgrad(d2, _ptr(dy, dlo * frame_out), _ptr(x, slo * frame_in), dw[kt])
            o.check(o.L.qt_unpack_conv3d_wgrad(_ptr(dw), _ptr(dW), self.cout, self.cin, self.cout_p, self.cin_p,
                                               1 if self.first else 0, _lib.stream_ptr()), "qt_unpack_conv3d_wgrad")
            return dW

        # The weight gradient hangs off the chain  dout -> dy -> dx -> (next block): on the weight-gradient stream (wside) its
        # MFMA kernels run beside the byte-moving BatchNorm / pooling passes of the NEXT block's backward (round 4; the 2-D
        # plan's arrangement, csrc/plan.hip).  x and dy were allocated on the compute stream: recorded on the side stream so
        # that the caching allocator does not hand their memory out again while it still reads them.
        if wside is not None:
            for t_ in ((x, y, dout, arg, dy) + tuple(stats) if raw and coef is not None else (x, dy)):
                t_.record_stream(wside.side)
            with wside.fork():
                dW = weight_gradient()
        else:
            dW = weight_gradient()
        dx = None
        if not self.first:   # data gradient: one 27-tap launch; conv3d_block2's on the slab-resident kernel (two passes)
            dx = torch.empty(rows, x.shape[1], dtype=dt, device=dev)   # (rows as wide as the input's: the previous block's pooled map)
            nscr = 0
            if slab:
                o.L.qt_conv3d_c32_dgrad_scratch_bytes.restype = _c.c_size_t
                nscr = int(o.L.qt_conv3d_c32_dgrad_scratch_bytes(B, T, H, W))
            if nscr:
                scr = torch.empty(nscr, dtype=torch.uint8, device=dev)
                o.conv3d_c32_dgrad(dt, dy, self.wd, dx, dx.shape[1], scr, nscr, B, T, H, W, flops=2.0 * rows * 27 * self.cin * self.cout,
                                   nbytes=esz * (rows * (self.cin + self.cout_p) + 27.0 * self.cin * self.cout_p))
            else:
                if dx.shape[1] != self.cin_p:
                    raise QtError(f"conv block: input rows of {dx.shape[1]} channels, the implicit GEMM writes {self.cin_p}")
                dd = self._desc(dt, _lib.QT_CONV_DGRAD, T, B, H, W)
                o.igemm(dd, _ptr(dy), _ptr(self.wd), _ptr(dx), flops=2.0 * rows * 27 * self.cin * self.cout,
                        nbytes=esz * (rows * (self.cin_p + self.cout_p) + 27.0 * self.cin_p * self.cout_p))
        return dx, dW, self._vec_grad(db), self._vec_grad(dgamma), self._vec_grad(dbeta)


def o_desc(dt, mode, images, h, w, k_per_tap, n_out, k, pad):
    return _Ops.conv_desc(dt, mode, images, h, w, k_per_tap, n_out, k, pad)


class _Lstm:
    """nn.LSTM(batch_first) layers on f32 [B][T][I]: input products by qt_gemm_small, recurrences by qt_lstm_*."""

    def __init__(self, lstm):
        self.m = lstm
        self.H, self.layers = lstm.hidden_size, lstm.num_layers

    def params(self, k):
        return (getattr(self.m, f"weight_ih_l{k}"), getattr(self.m, f"weight_hh_l{k}"),
                getattr(self.m, f"bias_ih_l{k}"), getattr(self.m, f"bias_hh_l{k}"))

    def forward(self, x, B, T, training, seed):
        o, dev, H = ops(), x.device, self.H
        f32 = dict(dtype=torch.float32, device=dev)
        self.saved = []
        inp = x
        for k in range(self.layers):
            w_ih, w_hh, b_ih, b_hh = (p.detach() for p in self.params(k))
            I = w_ih.shape[1]
            xproj = torch.empty(B * T, 4 * H, **f32)
            o.gemm(B * T, 4 * H, I, inp, I, 1, w_ih, I, 1, xproj, 4 * H, b_ih)
            whh_t = torch.empty(H, 4 * H, **f32)
            o.check(o.L.qt_transpose_f32(_ptr(w_hh), _ptr(whh_t), 4 * H, H, _lib.stream_ptr()), "qt_transpose_f32")
            gates, cell = torch.empty(B * T, 4 * H, **f32), torch.empty(B * T, H, **f32)
            hprev, hout = torch.empty(B * T, H, **f32), torch.empty(B * T, H, **f32)
            o.check(o.L.qt_lstm_forward(_ptr(xproj), _ptr(whh_t), _ptr(b_hh), _ptr(gates), _ptr(cell), _ptr(hprev), _ptr(hout),
                                        B, T, H, _lib.stream_ptr()), "qt_lstm_forward")
            nxt = hout
            p = self.m.dropout if (training and k + 1 < self.layers) else 0.0
            if p > 0:   # nn.LSTM drops the outputs of every layer but the last; the recurrent path keeps the undropped h
                nxt = hout.clone()
                o.dropout(nxt, B * T, H, H, 0, seed + 17 * (k + 1), p)
            self.saved.append((inp, gates, cell, hprev, nxt if p > 0 else None, p))
            inp = nxt
        return hout   # [B*T][H] of the last layer

    def backward(self, dlast, B, T):
        """dlast [B][H]: gradient of the last layer's final step -> parameter gradients in nn.LSTM order"""
        o, dev, H = ops(), dlast.device, self.H
        f32 = dict(dtype=torch.float32, device=dev)
        grads = [None] * (4 * self.layers)
        dh_all = None
        for k in reversed(range(self.layers)):
            w_ih, w_hh, _, _ = (p.detach() for p in self.params(k))
            inp, gates, cell, hprev, dropped, p = self.saved[k]
            I = w_ih.shape[1]
            dgates = torch.empty(B * T, 4 * H, **f32)
            o.check(o.L.qt_lstm_backward(_ptr(dh_all), _ptr(dlast) if k == self.layers - 1 else None, _ptr(gates), _ptr(cell),
                                         _ptr(w_hh), _ptr(dgates), B, T, H, _lib.stream_ptr()), "qt_lstm_backward")
            dW_ih, dW_hh = torch.empty(4 * H, I, **f32), torch.empty(4 * H, H, **f32)
            o.gemm(4 * H, I, B * T, dgates, 1, 4 * H, inp, 1, I, dW_ih, I)       # dgates^T x
            o.gemm(4 * H, H, B * T, dgates, 1, 4 * H, hprev, 1, H, dW_hh, H)     # dgates^T h_{t-1}
            db = torch.empty(4 * H, **f32)
            o.col_sum(torch.float32, dgates, B * T, 4 * H, 4 * H, db)
            grads[4 * k:4 * k + 4] = [dW_ih, dW_hh, db, db.clone()]
            if k > 0:
                dx = torch.empty(B * T, I, **f32)
                o.gemm(B * T, I, 4 * H, dgates, 4 * H, 1, w_ih, 1, I, dx, I)      # dgates W_ih
                pk = self.saved[k - 1][5]
                if pk > 0:   # through the inter-layer dropout: from the dropped activations themselves
                    o.check(o.L.qt_scale_by_nonzero(_ptr(dx), _ptr(self.saved[k - 1][4]), _c.c_longlong(B * T * I),
                                                    _c.c_float(1.0 / (1.0 - pk)), _lib.stream_ptr()), "qt_scale_by_nonzero")
                dh_all = dx
        return grads


class _ClipModel(nn.Module):
    """Shared driver: builds the block list / head description and runs them inside one autograd node."""

    def _init_clip_state(self, compute_dtype):
        self.compute_dtype = compute_dtype or default_compute_dtype()
        self._blocks = None

    def _check_inputs(self, image_sequence, numerical_sequence, need_numerical):
        if image_sequence.dim() != 5 or image_sequence.shape[2] != 3:
            raise ValueError(f"image_sequence must be [B,T,3,H,W], got {tuple(image_sequence.shape)}")
        if image_sequence.device.type != "cuda":
            raise QtError("this build of the model runs on an AMD GPU only: move the model and its inputs to cuda:N "
                          "(there is no CPU fallback for the product path)")
        B, T = int(image_sequence.shape[0]), int(image_sequence.shape[1])
        if need_numerical:
            if numerical_sequence is None or numerical_sequence.dim() != 3 or \
                    tuple(numerical_sequence.shape[:2]) != (B, T) or numerical_sequence.shape[2] != self.numerical_feature_dim:
                raise ValueError(f"numerical_sequence must be [{B},{T},{self.numerical_feature_dim}]")

    def _check_hooks(self):
        """The whole forward / backward is one autograd node that never calls the leaf modules: a hook registered on a
        submodule (the reference's Grad-CAM API targets `conv3d_final_features`, 3dcnn/models.py:181-182) would silently
        never fire.  Same policy as the 2-D models (quadtree.py::_check_hooks): raise instead of staying dead."""
        for name, mod in self.named_modules():
            if mod is self:
                continue
            if mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks:
                raise QtError(f"hook registered on submodule {name!r}: the clip models run as one fused autograd node and "
                              "serve no submodule hooks (hooks on the model itself work); they would never fire")

    def _cached_blocks(self, specs):
        """the conv blocks' executors (packed filter buffers) live as long as their modules do"""
        key = tuple((id(c), id(b), p, f) for c, b, p, f in specs)
        if self.__dict__.get("_blocks_key") != key:
            blocks = [_ConvBlock(c, b, p, f) for c, b, p, f in specs]
            for prev, nxt in zip(blocks, blocks[1:]):   # conv3d_block1 -> conv3d_block2 (32 -> 64): see _pooled_width
                prev.narrow_out = prev.first and bool(prev.pool_t) and nxt.cin == 32 and nxt.cout == 64
            self.__dict__["_blocks"] = blocks
            self.__dict__["_blocks_key"] = key
        return self.__dict__["_blocks"]

    def invalidate_packed(self):
        """Forget the packed copies of every conv block's filter / BatchNorm vectors: the next forward re-packs them.
        The cache (QTCNN_PACK_CACHE) follows the parameters' version counters, FusedAdam's raw updates and the model's own
        backward passes; a write that bumps none of them -- `p.data.copy_(ema)`, `p.data.mul_(...)`: `.data` carries its
        own version counter -- is invisible to it.  Call this after such a write (load_state_dict and in-place ops on the
        parameters themselves need nothing)."""
        for blk in self.__dict__.get("_blocks") or ():
            blk._packed_ver = None

    invalidate_packed_weights = invalidate_packed   # (the name the plan-backed 2-D models use, quadtree.py)

    def _run(self, image_sequence, numerical_sequence):
        self._check_hooks()
        params = [p for p in self.parameters()]
        # needs_input_grad reports requires_grad even under torch.no_grad(): without this an eval forward under no_grad kept
        # every activation and never took the fused eval kernels
        self._grad_mode = torch.is_grad_enabled()
        return _ClipFunction.apply(self, image_sequence, numerical_sequence, *params)


class _ClipFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, images, numerical, *params):
        # (grad mode is off inside Function.forward; `needs_input_grad` tells whether a backward may follow)
        keep = owner._grad_mode and any(ctx.needs_input_grad[3:])
        with torch.cuda.device(images.device):
            logits = owner._forward_impl(images, numerical, keep)
        ctx.owner = owner
        ctx.fwd_id = owner._fwd_counter
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        owner = ctx.owner
        if ctx.fwd_id != owner._fwd_counter:
            raise QtError("backward() after a later forward() on the same model: one set of activations is kept")
        with torch.cuda.device(dlogits.device):
            sync = getattr(owner, "_grad_sync", None)
            if sync is None:
                return (None, None, None, *owner._backward_impl(dlogits.contiguous().float()))
            # Data parallelism (dp.attach_data_parallel).  There is no plan behind the clip models, but their backward still
            # finishes its gradients in a known order: the dense head first, then the conv blocks from the last to the first.
            # _backward_impl hands them over in buckets as they become final (emit): each bucket is packed into one flat f32
            # tensor and its all-reduce starts on the communication stream while the earlier -- and by far heavier -- conv
            # blocks are still running; only the last, small bucket (blocks 4..1 + LSTM: < 7 MB of the 40 MB) is exposed.
            # (Round 3 reduced ONE bucket after the whole backward: the all-reduce was fully exposed.)  The gradients
            # returned to autograd are views of the averaged buckets: no copy back.
            buckets = []

            def emit(g, names, phase):
                live = [(n, g[n]) for n in names if g.get(n) is not None]
                if not live:
                    return
                flat = torch.cat([t.reshape(-1).float() for _, t in live])
                sync(flat, phase)
                buckets.append((live, flat))

            grads = owner._backward_impl(dlogits.contiguous().float(), emit)
            sync(None, 0)   # join: the compute stream sees the averaged buckets
            by_name = {}
            for live, flat in buckets:
                off = 0
                for n, t in live:
                    by_name[n] = flat[off:off + t.numel()].view_as(t).to(t.dtype)
                    off += t.numel()
            names = [n for n, _ in owner.named_parameters()]
            missing = [n for n, g0 in zip(names, grads) if g0 is not None and n not in by_name]
            if missing:
                raise QtError(f"data parallelism: gradients of {missing} were produced but never handed over in a bucket")
            grads = [None if g0 is None else by_name[n] for n, g0 in zip(names, grads)]
        return (None, None, None, *grads)


class _Head:
    """f32 dense tail shared by the two clip models: [image features | LSTM branch] -> classifier"""

    @staticmethod
    def linear_fwd(x, x_ld, x_off, lin, rows, out, out_ld, out_off, relu):
        ops().linear(x, x_ld, x_off, lin.weight.detach(), lin.bias.detach(), rows, out, out_ld, out_off, relu)

    @staticmethod
    def linear_bwd(dy, dy_ld, dy_off, x, x_ld, x_off, lin, rows, want_dx, dx=None, dx_ld=0, dx_off=0):
        o = ops()
        W = lin.weight.detach()
        N, K = W.shape
        dev = W.device
        dW = torch.empty(N, K, dtype=torch.float32, device=dev)
        db = torch.empty(N, dtype=torch.float32, device=dev)
        # dW[n][k] = sum_r dy[r][dy_off + n] * x[r][x_off + k]: A = dy^T (row stride 1, k stride dy_ld)
        d = _GemmDesc(N, K, rows, _lib.QT_F32, _lib.QT_F32, _lib.QT_F32, 1, dy_ld, 1, x_ld, K, 0, 0)
        o.check(o.L.qt_gemm_small(_c.byref(d), _ptr(dy, dy_off * 4), _ptr(x, x_off * 4), None, _ptr(dW), _lib.stream_ptr()),
                "qt_gemm_small")
        o.check(o.L.qt_col_sum(_lib.QT_F32, _ptr(dy, dy_off * 4), _c.c_longlong(rows), N, dy_ld, _ptr(db), 0, _lib.stream_ptr()),
                "qt_col_sum")
        if want_dx:   # dx[r][dx_off + k] = sum_n dy[r][dy_off + n] * W[n][k]
            d = _GemmDesc(rows, K, N, _lib.QT_F32, _lib.QT_F32, _lib.QT_F32, dy_ld, 1, 1, K, dx_ld, 0, 0)
            o.check(o.L.qt_gemm_small(_c.byref(d), _ptr(dy, dy_off * 4), _ptr(W), None, _ptr(dx, dx_off * 4), _lib.stream_ptr()),
                    "qt_gemm_small")
        return dW, db

    @staticmethod
    def relu_dropout_bwd(g, act, n, mul=1.0):
        """g = act > 0 ? g * mul : 0 (ReLU followed by dropout: `act` is the dropped output, mul = 1/(1-p))"""
        o = ops()
        o.check(o.L.qt_relu_mask_scale(_lib.QT_F32, _ptr(g), _ptr(act), _c.c_longlong(n), _c.c_float(mul), _lib.stream_ptr()),
                "qt_relu_mask_scale")


class Quadtree3DCNN(_ClipModel):
    """/root/reference/3dcnn/models.py:96-214."""

    def __init__(self, num_classes, sequence_length=8, cnn_3d_feature_dim=1024, numerical_feature_dim=47, dropout_rate=0.6,
                 mode='quadtree_3d_fusion', compute_dtype=None):
        super().__init__()
        self.mode = mode
        self.num_classes = num_classes
        self.sequence_length = sequence_length
        self.cnn_3d_feature_dim = cnn_3d_feature_dim
        self.numerical_feature_dim = numerical_feature_dim
        self.dropout_rate = dropout_rate
        if cnn_3d_feature_dim % 64:
            raise ValueError("cnn_3d_feature_dim must be a multiple of 64 for the gfx950 kernels (reference default 1024)")

        def block(cin, cout, pool):
            layers = [M.Conv3d(cin, cout, 3, padding=1), M.BatchNorm3d(cout), M.ReLU(inplace=True)]
            if pool:
                layers.append(M.MaxPool3d(pool, pool))
            return nn.Sequential(*layers)
        self.conv3d_block1 = block(3, 32, (1, 2, 2))
        self.conv3d_block2 = block(32, 64, (2, 2, 2))
        self.conv3d_block3 = block(64, 128, (2, 2, 2))
        self.conv3d_block4_new = block(128, 256, (1, 2, 2))
        self.conv3d_final_features = block(256, cnn_3d_feature_dim, None)
        self.global_avg_pool_3d = M.AdaptiveAvgPool3d((1, 1, 1))
        self.numerical_lstm = M.LSTM(numerical_feature_dim, numerical_feature_dim * 4, num_layers=2, batch_first=True,
                                     dropout=dropout_rate)
        self.numerical_lstm_output_dim = numerical_feature_dim * 4
        self.numerical_projection = nn.Sequential(M.Linear(self.numerical_lstm_output_dim, cnn_3d_feature_dim // 2),
                                                  M.ReLU(inplace=True), M.Dropout(dropout_rate))
        self.numerical_final_dim = cnn_3d_feature_dim // 2
        if mode == 'quadtree_3d_fusion':
            self.final_classifier_input_dim = cnn_3d_feature_dim + self.numerical_final_dim
        elif mode == 'quadtree_3d_image_only':
            self.final_classifier_input_dim = cnn_3d_feature_dim
        else:
            raise ValueError(f"Invalid mode for Quadtree3DCNN: {mode}. Choose from 'quadtree_3d_fusion', "
                             "'quadtree_3d_image_only'.")
        w = self.final_classifier_input_dim
        self.classifier = nn.Sequential(M.Linear(w, w // 2), M.ReLU(inplace=True), M.Dropout(dropout_rate),
                                        M.Linear(w // 2, num_classes))
        self.gradients = None
        self.activations = None
        self._fwd_counter = 0
        self._init_clip_state(compute_dtype)
        if self.numerical_lstm_output_dim not in (256, 188, 64):
            raise ValueError("the gfx950 LSTM kernel is instantiated for hidden sizes 256, 188 (= 4 x 47) and 64")

    def save_gradient_hook(self, module, grad_input, grad_output):
        self.gradients = grad_output[0]

    def save_activation_hook(self, module, input, output):
        self.activations = output

    def _conv_blocks(self):
        seqs = (self.conv3d_block1, self.conv3d_block2, self.conv3d_block3, self.conv3d_block4_new, self.conv3d_final_features)
        pools = (1, 2, 2, 1, 0)
        return self._cached_blocks([(s[0], s[1], p, i == 0) for i, (s, p) in enumerate(zip(seqs, pools))])

    def forward(self, image_sequence_input, numerical_sequence_input):
        fusion = self.mode == 'quadtree_3d_fusion'
        if fusion and numerical_sequence_input is not None:
            numerical_sequence_input = numerical_sequence_input.to(image_sequence_input.device)
        self._check_inputs(image_sequence_input, numerical_sequence_input, fusion)
        return self._run(image_sequence_input.contiguous().float(),
                         numerical_sequence_input.contiguous().float() if fusion else None)

    # ---- the graph --------------------------------------------------------------------------------------------
    def _forward_impl(self, images, numerical, keep):
        o, dev, dt = ops(), images.device, self.compute_dtype
        B, T, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[3]), int(images.shape[4])
        training = self.training
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0
        blocks = self._conv_blocks()
        fusion = self.mode == 'quadtree_3d_fusion'
        lstm = last = side = None
        if fusion:   # the numerical branch first: on the second stream it runs under the conv blocks
            lstm = _Lstm(self.numerical_lstm)
            side = _Side(dev) if LSTM_SIDE else None
            if side is not None:
                with side.fork():
                    hout = lstm.forward(numerical.view(B * T, -1), B, T, training, seed)
                    last = hout.view(B, T, lstm.H)[:, -1, :].contiguous()
            else:
                hout = lstm.forward(numerical.view(B * T, -1), B, T, training, seed)
                last = hout.view(B, T, lstm.H)[:, -1, :].contiguous()
        x = images   # conv3d_block1 reads the f32 clip itself (or packs it: _ConvBlock.forward)
        saved_blocks = []
        t, h, w = T, H, W
        for blk in blocks:
            blk.pack(dt, keep, self.__dict__.get("_bwd_count", 0))
            x, (t, h, w), sv = blk.forward(dt, x, t, B, h, w, training, keep)
            saved_blocks.append(sv)
        F_img = self.cnn_3d_feature_dim
        ld = self.final_classifier_input_dim
        fused = torch.empty(B, ld, dtype=torch.float32, device=dev)
        o.check(o.L.qt_avgpool_tb(_lib.qt_dtype(dt), _ptr(x), _ptr(fused), t, B, h * w, blocks[-1].cout_p, ld, 0,
                                  _lib.stream_ptr()), "qt_avgpool_tb")
        p = self.dropout_rate if training else 0.0
        if fusion:
            if side is not None:
                side.join(last)
            Hn = lstm.H
            _Head.linear_fwd(last, Hn, 0, self.numerical_projection[0], B, fused, ld, F_img, 1)
            if p > 0:
                o.dropout(fused, B, self.numerical_final_dim, ld, F_img, seed + 1, p)
        hid = torch.empty(B, ld // 2, dtype=torch.float32, device=dev)
        _Head.linear_fwd(fused, ld, 0, self.classifier[0], B, hid, ld // 2, 0, 1)
        if p > 0:
            o.dropout(hid, B, ld // 2, ld // 2, 0, seed + 2, p)
        logits = torch.empty(B, self.num_classes, dtype=torch.float32, device=dev)
        _Head.linear_fwd(hid, ld // 2, 0, self.classifier[3], B, logits, self.num_classes, 0, 0)
        self._fwd_counter += 1
        self._saved = (blocks, saved_blocks, (t, h, w, B, T), fused, hid, lstm, last, p) if keep else None
        return logits

    def _backward_impl(self, dlogits, emit=None):
        """emit(g, names, phase): data-parallel hand-over of the gradients `names` of dict g, final as of this point"""
        self.__dict__["_bwd_count"] = self.__dict__.get("_bwd_count", 0) + 1   # gradients exist: packed weights may go stale
        o, dt = ops(), self.compute_dtype
        blocks, saved_blocks, (t, h, w, B, T), fused, hid, lstm, last, p = self._saved
        dev = dlogits.device
        ld, F_img = self.final_classifier_input_dim, self.cnn_3d_feature_dim
        mul = 1.0 / (1.0 - p) if p > 0 else 1.0
        g = {}
        dhid = torch.empty(B, ld // 2, dtype=torch.float32, device=dev)
        g["classifier.3.weight"], g["classifier.3.bias"] = _Head.linear_bwd(
            dlogits, self.num_classes, 0, hid, ld // 2, 0, self.classifier[3], B, True, dhid, ld // 2, 0)
        _Head.relu_dropout_bwd(dhid, hid, B * (ld // 2), mul)
        dfused = torch.empty(B, ld, dtype=torch.float32, device=dev)
        g["classifier.0.weight"], g["classifier.0.bias"] = _Head.linear_bwd(
            dhid, ld // 2, 0, fused, ld, 0, self.classifier[0], B, True, dfused, ld, 0)
        if lstm is not None:
            Hn, Fn = lstm.H, self.numerical_final_dim
            dproj = dfused[:, F_img:].contiguous()
            _Head.relu_dropout_bwd(dproj, fused[:, F_img:].contiguous(), B * Fn, mul)
            dlast = torch.empty(B, Hn, dtype=torch.float32, device=dev)
            g["numerical_projection.0.weight"], g["numerical_projection.0.bias"] = _Head.linear_bwd(
                dproj, Fn, 0, last, Hn, 0, self.numerical_projection[0], B, True, dlast, Hn, 0)
            side = _Side(dev) if LSTM_SIDE else None
            if side is not None:   # (joined after the image branch's backward)
                with side.fork():
                    lg = lstm.backward(dlast, B, T)
            else:
                lg = lstm.backward(dlast, B, T)
            for k in range(lstm.layers):
                for j, nm in enumerate(("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
                    g[f"numerical_lstm.{nm}_l{k}"] = lg[4 * k + j]
        else:
            side = None
        if emit is not None:   # bucket 1: the dense head (the LSTM's gradients are still running on the side stream)
            emit(g, [k for k in g if k.startswith(("classifier.", "numerical_projection."))], 1)
        # image branch
        C_last = blocks[-1].cout_p
        dout = torch.empty(t * B * h * w, C_last, dtype=dt, device=dev)
        o.check(o.L.qt_avgpool_tb_bwd(_lib.qt_dtype(dt), _ptr(dfused), _ptr(dout), t, B, h * w, C_last, ld, 0, _lib.stream_ptr()),
                "qt_avgpool_tb_bwd")
        names = ("conv3d_block1", "conv3d_block2", "conv3d_block3", "conv3d_block4_new", "conv3d_final_features")
        wside = _Side(dev, 1) if WGRAD_SIDE else None
        for blk, sv, nm in zip(reversed(blocks), reversed(saved_blocks), reversed(names)):
            dout, dW, db, dgamma, dbeta = blk.backward(dt, dout, sv, wside)
            g[f"{nm}.0.weight"], g[f"{nm}.0.bias"], g[f"{nm}.1.weight"], g[f"{nm}.1.bias"] = dW, db, dgamma, dbeta
            if emit is not None and nm == "conv3d_final_features":   # bucket 2: 7.1 M of the 9.9 M parameters, ready first
                if wside is not None:
                    wside.join(dW)   # (the bucket is packed on the compute stream)
                emit(g, [k for k in g if k.startswith(nm + ".")], 2)
        if wside is not None:
            wside.join(*[g[f"{nm}.0.weight"] for nm in names])
        if side is not None:
            side.join(*lg)
        if emit is not None:   # bucket 3: conv blocks 4 .. 1 and the LSTM (joined above)
            emit(g, [k for k in g if k.startswith(("conv3d_block", "numerical_lstm."))], 4)
        self._saved = None
        return [g.get(n) for n, _ in self.named_parameters()]


class Ji3DCNN(_ClipModel):
    """/root/reference/cnn+lstm/models.py:93-142."""

    def __init__(self, num_classes, sequence_length=4, numerical_feature_dim=47, dropout_rate=0.5, compute_dtype=None):
        super().__init__()
        self.num_classes = num_classes
        self.sequence_length = sequence_length
        self.numerical_feature_dim = numerical_feature_dim
        self.dropout_rate = dropout_rate

        def conv_3d_block(cin, cout):
            return nn.Sequential(M.Conv3d(cin, cout, 3, padding=1), M.BatchNorm3d(cout), M.ReLU(inplace=True))
        self.visual_stream = nn.Sequential(
            conv_3d_block(3, 32), M.MaxPool3d((1, 2, 2)), conv_3d_block(32, 64), M.MaxPool3d((2, 2, 2)),
            conv_3d_block(64, 128), M.AdaptiveAvgPool3d((1, 1, 1)))
        self.numerical_lstm = M.LSTM(numerical_feature_dim, 64, num_layers=1, batch_first=True)
        self.classifier = nn.Sequential(M.Linear(128 + 64, 128), M.ReLU(), M.Dropout(dropout_rate), M.Linear(128, num_classes))
        self._fwd_counter = 0
        self._init_clip_state(compute_dtype)

    def _conv_blocks(self):
        vs = self.visual_stream
        return self._cached_blocks([(vs[0][0], vs[0][1], 1, True), (vs[2][0], vs[2][1], 2, False), (vs[4][0], vs[4][1], 0, False)])

    def forward(self, image_sequence, numerical_sequence):
        numerical_sequence = numerical_sequence.to(image_sequence.device)
        self._check_inputs(image_sequence, numerical_sequence, True)
        return self._run(image_sequence.contiguous().float(), numerical_sequence.contiguous().float())

    def _forward_impl(self, images, numerical, keep):
        o, dev, dt = ops(), images.device, self.compute_dtype
        B, T, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[3]), int(images.shape[4])
        training = self.training
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0
        blocks = self._conv_blocks()
        x = images   # conv3d_block1 reads the f32 clip itself (or packs it: _ConvBlock.forward)
        saved_blocks = []
        t, h, w = T, H, W
        for blk in blocks:
            blk.pack(dt, keep, self.__dict__.get("_bwd_count", 0))
            x, (t, h, w), sv = blk.forward(dt, x, t, B, h, w, training, keep)
            saved_blocks.append(sv)
        ld = 128 + 64
        fused = torch.empty(B, ld, dtype=torch.float32, device=dev)
        o.check(o.L.qt_avgpool_tb(_lib.qt_dtype(dt), _ptr(x), _ptr(fused), t, B, h * w, blocks[-1].cout_p, ld, 0,
                                  _lib.stream_ptr()), "qt_avgpool_tb")
        lstm = _Lstm(self.numerical_lstm)
        hout = lstm.forward(numerical.view(B * T, -1), B, T, training, seed)
        fused[:, 128:].copy_(hout.view(B, T, 64)[:, -1, :])      # torch.cat((v_out, n_out), dim=1)
        p = self.dropout_rate if training else 0.0
        hid = torch.empty(B, 128, dtype=torch.float32, device=dev)
        _Head.linear_fwd(fused, ld, 0, self.classifier[0], B, hid, 128, 0, 1)
        if p > 0:
            o.dropout(hid, B, 128, 128, 0, seed + 2, p)
        logits = torch.empty(B, self.num_classes, dtype=torch.float32, device=dev)
        _Head.linear_fwd(hid, 128, 0, self.classifier[3], B, logits, self.num_classes, 0, 0)
        self._fwd_counter += 1
        self._saved = (blocks, saved_blocks, (t, h, w, B, T), fused, hid, lstm, p) if keep else None
        return logits

    def _backward_impl(self, dlogits, emit=None):
        self.__dict__["_bwd_count"] = self.__dict__.get("_bwd_count", 0) + 1   # gradients exist: packed weights may go stale
        o, dt = ops(), self.compute_dtype
        blocks, saved_blocks, (t, h, w, B, T), fused, hid, lstm, p = self._saved
        dev = dlogits.device
        ld = 128 + 64
        mul = 1.0 / (1.0 - p) if p > 0 else 1.0
        g = {}
        dhid = torch.empty(B, 128, dtype=torch.float32, device=dev)
        g["classifier.3.weight"], g["classifier.3.bias"] = _Head.linear_bwd(
            dlogits, self.num_classes, 0, hid, 128, 0, self.classifier[3], B, True, dhid, 128, 0)
        _Head.relu_dropout_bwd(dhid, hid, B * 128, mul)
        dfused = torch.empty(B, ld, dtype=torch.float32, device=dev)
        g["classifier.0.weight"], g["classifier.0.bias"] = _Head.linear_bwd(
            dhid, 128, 0, fused, ld, 0, self.classifier[0], B, True, dfused, ld, 0)
        lg = lstm.backward(dfused[:, 128:].contiguous(), B, T)
        for j, nm in enumerate(("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
            g[f"numerical_lstm.{nm}_l0"] = lg[j]
        if emit is not None:   # bucket 1: classifier + LSTM (this model's LSTM runs on the compute stream)
            emit(g, list(g), 1)
        C_last = blocks[-1].cout_p
        dout = torch.empty(t * B * h * w, C_last, dtype=dt, device=dev)
        o.check(o.L.qt_avgpool_tb_bwd(_lib.qt_dtype(dt), _ptr(dfused), _ptr(dout), t, B, h * w, C_last, ld, 0, _lib.stream_ptr()),
                "qt_avgpool_tb_bwd")
        wside = _Side(dev, 1) if WGRAD_SIDE else None
        for blk, sv, nm in zip(reversed(blocks), reversed(saved_blocks), ("visual_stream.4", "visual_stream.2", "visual_stream.0")):
            dout, dW, db, dgamma, dbeta = blk.backward(dt, dout, sv, wside)
            g[f"{nm}.0.weight"], g[f"{nm}.0.bias"], g[f"{nm}.1.weight"], g[f"{nm}.1.bias"] = dW, db, dgamma, dbeta
        if wside is not None:
            wside.join(*[g[f"{nm}.0.weight"] for nm in ("visual_stream.4", "visual_stream.2", "visual_stream.0")])
        if emit is not None:   # bucket 2: the three conv blocks (0.3 M parameters)
            emit(g, [k for k in g if k.startswith("visual_stream.")], 2)
        self._saved = None
        return [g.get(n) for n, _ in self.named_parameters()]
