"""Host side of the plan executor: binds a model's tensors to a `qt_plan`
(include/qtcnn.h) and exposes it to autograd as ONE differentiable function
(image, numerical, *parameters) -> logits.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every
FLOP of the hot path is issued by libqtcnn_hip.so.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import QtError

QT_MODEL_QUADTREE, QT_MODEL_STANDARD_RESNET, QT_MODEL_ATTENTION, QT_MODEL_CNN_LSTM = 0, 1, 2, 3
MODES = {"fusion": 0, "image_only": 1, "numerical_only": 2}
QT_BWD_HEAD, QT_BWD_LAYER4, QT_BWD_LAYER32, QT_BWD_LAYER1, QT_BWD_ALL = 1, 2, 4, 8, 15


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", ctypes.c_int), ("batch", ctypes.c_int), ("num_classes", ctypes.c_int),
        ("model", ctypes.c_int), ("mode", ctypes.c_int), ("numerical_dim", ctypes.c_int),
        ("dropout_p", ctypes.c_float), ("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float),
        ("seq_len", ctypes.c_int), ("lstm_hidden", ctypes.c_int),
    ]


def default_compute_dtype():
    """bf16 is the throughput build; QTCNN_DTYPE=f32 selects the exact-f32 MFMA
    parity build of the same kernels."""
    name = os.environ.get("QTCNN_DTYPE", "bf16").lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    raise QtError(f"QTCNN_DTYPE={name!r}: expected bf16 or f32")


class AdamDesc(ctypes.Structure):   # qt_adam_desc
    _fields_ = [("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("weight_decay", ctypes.c_float), ("grad_scale", ctypes.c_float), ("step", ctypes.c_int)]


class AdamItem(ctypes.Structure):   # qt_adam_item
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("numel", ctypes.c_longlong)]


def _bind_api(L):
    if getattr(L, "_plan_bound", False):
        return
    L.qt_plan_create.argtypes = [ctypes.POINTER(PlanDesc), ctypes.POINTER(ctypes.c_void_p)]
    L.qt_plan_destroy.argtypes = [ctypes.c_void_p]
    L.qt_plan_destroy.restype = None
    L.qt_plan_num_tensors.argtypes = [ctypes.c_void_p]
    L.qt_plan_tensor_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.qt_plan_tensor_name.restype = ctypes.c_char_p
    L.qt_plan_tensor_kind.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.qt_plan_tensor_shape.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.qt_plan_workspace_bytes.argtypes = [ctypes.c_void_p]
    L.qt_plan_workspace_bytes.restype = ctypes.c_size_t
    L.qt_plan_init_workspace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.qt_plan_pack_weights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.qt_plan_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p]
    L.qt_plan_backward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.qt_plan_side_fence.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.qt_plan_adam_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.POINTER(AdamDesc), ctypes.c_int, ctypes.c_void_p]
    L.qt_plan_find_buffer.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    L._plan_bound = True


class PlanEngine:
    """One qt_plan + its workspace for one (model variant, device, dtype, max batch)."""

    def __init__(self, model_kind, mode, num_classes, numerical_dim, dropout_p, batch, dtype, device, seq_len=0,
                 lstm_hidden=0):
        self.L = _lib.lib()
        _bind_api(self.L)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise QtError("the QuadtreeCNN plan runs on an AMD GPU only (device must be cuda:N); no CPU fallback")
        self.dtype = dtype
        self.max_batch = int(batch)
        self.num_classes = num_classes
        if model_kind == QT_MODEL_STANDARD_RESNET:
            self.fused_ld = 512
        elif model_kind == QT_MODEL_ATTENTION:
            self.fused_ld = 512 + 4 * 128 + 64 + 128
        elif model_kind == QT_MODEL_CNN_LSTM:
            self.fused_ld = 512 + 128
        else:
            self.fused_ld = {"fusion": 5376, "image_only": 5120, "numerical_only": 256}.get(mode, 5376)
        self.rows_per_logit = int(seq_len) if model_kind == QT_MODEL_CNN_LSTM else 1
        desc = PlanDesc(_lib.qt_dtype(dtype), self.max_batch, num_classes, model_kind, MODES.get(mode, 0),
                        numerical_dim, float(dropout_p), 1e-5, 0.1, int(seq_len), int(lstm_hidden))
        handle = ctypes.c_void_p()
        _lib.check(self.L.qt_plan_create(ctypes.byref(desc), ctypes.byref(handle)), "qt_plan_create")
        self.handle = handle
        n = self.L.qt_plan_num_tensors(handle)
        self.names, self.kinds, self.shapes = [], [], []
        dims = (ctypes.c_int * 4)()
        for i in range(n):
            self.names.append(self.L.qt_plan_tensor_name(handle, i).decode())
            self.kinds.append(self.L.qt_plan_tensor_kind(handle, i))
            nd = self.L.qt_plan_tensor_shape(handle, i, dims)
            self.shapes.append(tuple(dims[k] for k in range(nd)))
        self.index = {name: i for i, name in enumerate(self.names)}
        nbytes = self.L.qt_plan_workspace_bytes(handle)
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self.workspace.data_ptr()) % 256
            self.ws_ptr = ctypes.c_void_p(self.workspace.data_ptr() + off)
            _lib.check(self.L.qt_plan_init_workspace(handle, self.ws_ptr, _lib.stream_ptr()), "qt_plan_init_workspace")
        self.workspace_bytes = nbytes
        self._tensor_ptrs = (ctypes.c_void_p * n)()
        self._packed_version = None
        # parameter gradients were handed out since the last pack: an optimizer step follows, and fused /
        # foreach optimizers update parameters without bumping Tensor._version
        self._weights_stale = True
        self.fwd_counter = 0
        self.grad_sync = None  # optional callable(bucket: Tensor, phase: int) for data parallelism
        self._flat_grad = None       # flat f32 gradient buffer of the last backward (re-used when nobody holds its views)

    @staticmethod
    def _storage_idle(t):
        """True when no tensor but `t` itself shares t's storage (the caller dropped every gradient view)."""
        try:
            return torch._C._storage_Use_Count(t.untyped_storage()._cdata) <= 2  # t + the temporary handle
        except Exception:
            return False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.L.qt_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def buffer(self, name, shape):
        """View of a named workspace buffer as a tensor of the compute dtype (debug / tests)."""
        off = ctypes.c_size_t()
        _lib.check(self.L.qt_plan_find_buffer(self.handle, name.encode(), ctypes.byref(off)), "qt_plan_find_buffer")
        base = (self.ws_ptr.value - self.workspace.data_ptr()) + off.value
        n = int(torch.Size(shape).numel()) * (2 if self.dtype == torch.bfloat16 else 4)
        return self.workspace[base:base + n].view(self.dtype).view(shape)

    def side_fence(self, torch_stream):
        """Make `torch_stream` wait for the plan's weight-gradient stream (see qt_plan_side_fence)."""
        _lib.check(self.L.qt_plan_side_fence(self.handle, ctypes.c_void_p(torch_stream.cuda_stream)),
                   "qt_plan_side_fence")

    def buffer_ld(self, name):
        """Row length (elements) of the fused feature matrix / its gradient."""
        if name in ("fused", "dfused"):
            return self.fused_ld
        raise QtError(f"no leading dimension known for {name}")

    # -- binding --------------------------------------------------------------
    def bind(self, tensors):
        """tensors: dict state_dict-key -> tensor living on self.device."""
        for i, name in enumerate(self.names):
            t = tensors.get(name)
            if t is None:
                raise QtError(f"model is missing tensor {name!r} required by the plan")
            if tuple(t.shape) != self.shapes[i]:
                raise QtError(f"{name}: shape {tuple(t.shape)} != plan shape {self.shapes[i]}")
            want = torch.int64 if self.kinds[i] == 2 else torch.float32
            if t.dtype != want or t.device != self.device or not t.is_contiguous():
                raise QtError(f"{name}: expected contiguous {want} on {self.device}, got {t.dtype} on {t.device}")
            ptr = t.data_ptr()
            if self._tensor_ptrs[i] != ptr:
                self._weights_stale = True      # a parameter was replaced (load / .to() / new storage)
            self._tensor_ptrs[i] = ptr

    def adam_step(self, by_index, desc):
        """Fused optimizer step + operand re-packing (qt_plan_adam_step).  by_index: plan tensor
        index -> (grad, exp_avg, exp_avg_sq) f32 tensors on this device; the parameters themselves
        are the tensors bound by the last forward."""
        n = len(self.names)
        g, m, v = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        for idx, (gt, mt, vt) in by_index.items():
            for t in (gt, mt, vt):
                if t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous():
                    raise QtError(f"{self.names[idx]}: gradient / Adam state must be contiguous f32 on {self.device}")
            g[idx], m[idx], v[idx] = gt.data_ptr(), mt.data_ptr(), vt.data_ptr()
        _lib.check(self.L.qt_plan_adam_step(self.handle, self.ws_ptr, self._tensor_ptrs, g, m, v, ctypes.byref(desc), 1,
                                            _lib.stream_ptr()), "qt_plan_adam_step")
        self._weights_stale = False   # the step re-packed every operand copy from the updated masters
        if self._packed_version is not None:
            self._packed_version = (self._packed_version[0], True)

    def invalidate_weights(self):
        """The f32 master weights changed behind the model's back: re-pack them at the next forward."""
        self._weights_stale = True

    def pack_weights(self, version, for_backward):
        """Re-pack the compute-dtype operand copies when the f32 masters may have changed: the
        parameters' version counters moved, a parameter's storage was replaced, or gradients were
        produced since the last pack (torch's fused Adam/SGD kernels update parameters WITHOUT
        bumping Tensor._version, so the counter alone would leave the plan on stale weights)."""
        key = (version, bool(for_backward))
        if not self._weights_stale and self._packed_version is not None and self._packed_version[0] == version and \
                (self._packed_version[1] or not for_backward):
            return
        _lib.check(self.L.qt_plan_pack_weights(self.handle, self.ws_ptr, self._tensor_ptrs, int(for_backward),
                                               _lib.stream_ptr()), "qt_plan_pack_weights")
        self._packed_version = key
        self._weights_stale = False

    # -- execution --------------------------------------------------------------
    def forward(self, image, numerical, training, seed):
        batch = int(image.shape[0]) if image is not None else int(numerical.shape[0])
        logits = torch.empty(batch // self.rows_per_logit, self.num_classes, dtype=torch.float32, device=self.device)
        _lib.check(self.L.qt_plan_forward(self.handle, self.ws_ptr, self._tensor_ptrs,
                                          _lib.ptr(image), _lib.ptr(numerical), _lib.ptr(logits), batch,
                                          int(training), ctypes.c_ulonglong(seed), _lib.stream_ptr()),
                   "qt_plan_forward")
        self.fwd_counter += 1
        return logits

    @staticmethod
    def gradient_buckets(names, wanted):
        """Layout of the flat f32 gradient buffer: `wanted` = [(plan tensor index, shape)] -> (offset of every tensor, its
        element count, the END offset of each of the four phase buckets [head | layer4 | layer3 + layer2 | layer1 + stem],
        total elements).  Pure host arithmetic (tests/test_boundary_cpu.py pins the bucket sizes of SURVEY.md 8e with it)."""
        def bucket_of(idx):
            name = names[idx]
            if not name.startswith("base_cnn."):
                return 0
            if name.startswith("base_cnn.layer4."):
                return 1
            return 2 if name.startswith(("base_cnn.layer3.", "base_cnn.layer2.")) else 3
        order = sorted(wanted, key=lambda w: bucket_of(w[0]))  # stable: keeps parameter order inside a bucket
        sizes = {idx: int(torch.Size(shape).numel()) for idx, shape in wanted}
        # keep every view 16-byte aligned
        offs, total, ends = {}, 0, [0, 0, 0, 0]
        for idx, _ in order:
            offs[idx] = total
            total += (sizes[idx] + 3) // 4 * 4
            ends[bucket_of(idx)] = total
        for b in range(1, 4):
            ends[b] = max(ends[b], ends[b - 1])
        return offs, sizes, ends, total

    def backward(self, dlogits, numerical, wanted):
        """wanted: list of (plan tensor index, shape) in the order gradients are
        returned.  Gradients are views of one flat f32 buffer laid out
        [head | layer4 | layer3 + layer2 | layer1 + stem]: the order in which backward finishes them, so a
        data-parallel caller all-reduces four buckets (59 / 34 / 10.5 / 0.6 MB), each while the next phase
        runs; only the last, smallest one is exposed."""
        offs, sizes, ends, total = self.gradient_buckets(self.names, wanted)
        # One flat buffer per (layout): re-used across steps unless a previous step's gradient views are still
        # referenced by the caller (`.grad` kept with zero_grad(set_to_none=False), accumulation, retained graphs):
        # then that step's buffer stays theirs and a fresh one is taken.
        flat = self._flat_grad
        if flat is None or flat.numel() != max(total, 1) or not self._storage_idle(flat):
            flat = torch.empty(max(total, 1), dtype=torch.float32, device=self.device)
            self._flat_grad = flat
        grad_ptrs = (ctypes.c_void_p * len(self.names))()
        views = {}
        for idx, shape in wanted:
            v = flat[offs[idx]:offs[idx] + sizes[idx]].view(shape)
            views[idx] = v
            grad_ptrs[idx] = v.data_ptr()

        if wanted:
            self._weights_stale = True

        def run(phase):
            _lib.check(self.L.qt_plan_backward(self.handle, self.ws_ptr, self._tensor_ptrs, grad_ptrs,
                                               _lib.ptr(numerical), _lib.ptr(dlogits), phase, _lib.stream_ptr()),
                       "qt_plan_backward")

        if self.grad_sync is None:
            run(QT_BWD_ALL)
        else:
            begin = 0
            for b, phase in enumerate((QT_BWD_HEAD, QT_BWD_LAYER4, QT_BWD_LAYER32, QT_BWD_LAYER1)):
                run(phase)
                if ends[b] > begin:
                    self.grad_sync(flat[begin:ends[b]], phase, self.side_fence)
                begin = ends[b]
            self.grad_sync(None, 0, None)  # join
        return [views[idx] for idx, _ in wanted]


class PlanFunction(torch.autograd.Function):
    """(image, numerical, *parameters) -> logits as a single autograd node."""

    @staticmethod
    def forward(ctx, owner, image, numerical, *params):
        engine = owner._engine
        # 1 train; 0 eval (fused epilogues, nothing kept); 2 eval statistics with everything kept for a backward through
        # the backbone (model.eval() + logits.backward() with trainable backbone parameters: Grad-CAM on the reference's
        # all-trainable variant, Quadtree_from scratch/grad_cam.py:72-83)
        # (decided in _PlanModel._run: grad mode is switched off inside Function.forward)
        training = 1 if owner.training else (2 if owner._eval_keep_for_backward else 0)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training == 1 else 0
        logits = engine.forward(image, numerical, training, seed)
        ctx.owner = owner
        ctx.engine = engine
        ctx.fwd_id = engine.fwd_counter
        ctx.numerical = numerical
        ctx.param_index = owner._param_plan_index
        ctx.param_shapes = [tuple(p.shape) for p in params]
        owner._fire_layer4_forward_hooks(engine, logits.shape[0])
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        engine = ctx.engine
        if ctx.fwd_id != engine.fwd_counter:
            raise QtError("backward() after a later forward() on the same model: the plan keeps one set of "
                          "activations; call backward before the next forward")
        needs = ctx.needs_input_grad[3:]
        wanted = [(ctx.param_index[i], ctx.param_shapes[i]) for i, need in enumerate(needs)
                  if need and ctx.param_index[i] >= 0]
        grads = engine.backward(dlogits.contiguous().float(), ctx.numerical, wanted)
        ctx.owner._fire_layer4_backward_hooks(engine, dlogits.shape[0])
        out, k = [], 0
        for i, need in enumerate(needs):
            if need and ctx.param_index[i] >= 0:
                out.append(grads[k])
                k += 1
            else:
                out.append(None)
        return (None, None, None, *out)
