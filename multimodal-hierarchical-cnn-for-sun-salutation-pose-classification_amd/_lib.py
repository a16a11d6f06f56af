"""ctypes binding of libqtcnn_hip.so (the C ABI declared in include/qtcnn.h).

There is no CPU or PyTorch fallback: if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C <pkg>/csrc`)
every product entry point raises.
"""
import ctypes
import os

# Side stream + RCCL streams need more than ROCm's default 4 hardware queues to really overlap
# (see bench.py); only effective if the HIP runtime has not been initialised yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: F401,E402  (loads the process-wide HIP runtime first)

_HERE = os.path.dirname(os.path.abspath(__file__))
# QTCNN_LIB_PATH: another build of the same ABI (A/B measurements of two kernel versions on one box)
LIB_PATH = os.environ.get("QTCNN_LIB_PATH") or os.path.join(_HERE, "libqtcnn_hip.so")

QT_F32, QT_BF16 = 0, 1
QT_CONV_FWD, QT_CONV_DGRAD = 0, 1


class QtError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", ctypes.c_int), ("mode", ctypes.c_int), ("batch", ctypes.c_int),
        ("in_h", ctypes.c_int), ("in_w", ctypes.c_int),
        ("out_h", ctypes.c_int), ("out_w", ctypes.c_int),
        ("k_per_tap", ctypes.c_int), ("n_out", ctypes.c_int),
        ("kh", ctypes.c_int), ("kw", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
        ("src_img_stride", ctypes.c_longlong),
        ("src_row_stride", ctypes.c_int), ("src_pix_stride", ctypes.c_int),
        ("quad", ctypes.c_int), ("relu", ctypes.c_int),
        ("dst_sub", ctypes.c_int), ("dst_h", ctypes.c_int), ("dst_w", ctypes.c_int),
        ("dst_off_h", ctypes.c_int), ("dst_off_w", ctypes.c_int),
        ("dst_merge", ctypes.c_int), ("dst_merge_res0", ctypes.c_int),
        ("kt", ctypes.c_int), ("frames", ctypes.c_int),
        ("dst_merge_extra", ctypes.c_int),   # with dst_merge: a fifth tap slot read from ConvIO.extra_src (class (0,0) only)
    ]


class ConvIO(ctypes.Structure):
    _fields_ = [
        ("src", ctypes.c_void_p), ("weight", ctypes.c_void_p), ("dst", ctypes.c_void_p),
        ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("residual", ctypes.c_void_p), ("relu_mask", ctypes.c_void_p),
        ("stats_partial", ctypes.c_void_p),
        # bwd_bn[2]: {y, mean, invstd, partial}
        ("bn0_y", ctypes.c_void_p), ("bn0_mean", ctypes.c_void_p), ("bn0_invstd", ctypes.c_void_p),
        ("bn0_partial", ctypes.c_void_p),
        ("bn1_y", ctypes.c_void_p), ("bn1_mean", ctypes.c_void_p), ("bn1_invstd", ctypes.c_void_p),
        ("bn1_partial", ctypes.c_void_p),
        ("relu_mask_bits", ctypes.c_void_p),   # [M][n_out/8] bytes: the ReLU mask as one bit per element (or NULL)
        ("extra_src", ctypes.c_void_p),        # second gradient map of the fifth tap slot (dst_merge_extra) or NULL
    ]


class ConvS2Desc(ctypes.Structure):   # qt_conv_s2_desc
    _fields_ = [("dtype", ctypes.c_int), ("batch", ctypes.c_int), ("in_h", ctypes.c_int), ("in_w", ctypes.c_int),
                ("c_in", ctypes.c_int), ("c_out", ctypes.c_int), ("relu_conv", ctypes.c_int), ("relu_down", ctypes.c_int)]


class ConvS2IO(ctypes.Structure):     # qt_conv_s2_io
    _fields_ = [("src", ctypes.c_void_p), ("w_conv", ctypes.c_void_p), ("w_down", ctypes.c_void_p),
                ("y_conv", ctypes.c_void_p), ("y_down", ctypes.c_void_p),
                ("scale_conv", ctypes.c_void_p), ("shift_conv", ctypes.c_void_p),
                ("scale_down", ctypes.c_void_p), ("shift_down", ctypes.c_void_p),
                ("stats_conv", ctypes.c_void_p), ("stats_down", ctypes.c_void_p)]


_lib = None


def lib():
    """The loaded library; raises QtError (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QtError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the product path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.qt_last_error.restype = ctypes.c_char_p
        _lib.qt_version.restype = ctypes.c_int
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().qt_last_error().decode(errors="replace")
        raise QtError(f"{what} failed with status {status}: {msg}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def qt_dtype(torch_dtype):
    if torch_dtype == torch.float32:
        return QT_F32
    if torch_dtype == torch.bfloat16:
        return QT_BF16
    raise QtError(f"unsupported activation dtype {torch_dtype}")
