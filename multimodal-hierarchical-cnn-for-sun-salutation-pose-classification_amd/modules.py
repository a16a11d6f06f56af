"""Parameter-holding module tree with the reference's names.

The reference builds its models out of torchvision's ResNet-18 and stock
torch.nn layers (/root/reference/Quadtree_from scratch/models.py:221-271,
/root/reference/resnet/models.py:12-41,76-129).  The drop-in keeps the same
attribute tree -- so `state_dict()` has the reference's 252 / 246 keys, Grad-CAM
code can find `model.base_cnn.layer4`, optimizers see the same 72 parameters in
the same order -- but the leaves only HOLD tensors: all arithmetic runs in the
HIP plan executor behind the owning model's forward().  Calling a leaf directly
raises; there is no torch / CPU fallback path.
"""
import math
import os

import torch
import torch.nn as nn

from ._lib import QtError


class _Leaf(nn.Module):
    def forward(self, *args, **kwargs):
        raise QtError(
            f"{type(self).__name__} is a parameter holder of the MI355X QuadtreeCNN build; "
            "call the owning model (its forward runs the HIP plan). There is no per-module fallback.")


class Conv2d(_Leaf):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size, kernel_size), (stride, stride), (padding, padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):  # torch.nn.Conv2d defaults
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, padding={self.padding}, bias={self.bias is not None}")


class BatchNorm2d(_Leaf):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


class Conv3d(_Leaf):
    """nn.Conv3d holder: weight [O][I][kT][kH][kW], bias [O]; torch's default initialisation"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = k, (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        self.padding = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_channels * k[0] * k[1] * k[2])
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}")


class BatchNorm3d(BatchNorm2d):
    pass


class MaxPool3d(_Leaf):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride or kernel_size, padding


class AdaptiveAvgPool3d(_Leaf):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size


class Linear(_Leaf):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}"


class LSTM(_Leaf):
    """Parameter holder with torch.nn.LSTM's names and order: weight_ih_l{k}, weight_hh_l{k}, bias_ih_l{k},
    bias_hh_l{k} per layer; gate rows i, f, g, o; uniform(-1/sqrt(hidden), 1/sqrt(hidden)) initialisation."""

    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=False, dropout=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.batch_first, self.dropout = batch_first, dropout
        bound = 1.0 / math.sqrt(hidden_size)
        for k in range(num_layers):
            shapes = (("weight_ih", (4 * hidden_size, input_size if k == 0 else hidden_size)),
                      ("weight_hh", (4 * hidden_size, hidden_size)), ("bias_ih", (4 * hidden_size,)),
                      ("bias_hh", (4 * hidden_size,)))
            for name, shape in shapes:
                p = nn.Parameter(torch.empty(*shape))
                nn.init.uniform_(p, -bound, bound)
                setattr(self, f"{name}_l{k}", p)

    def extra_repr(self):
        return f"{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, batch_first={self.batch_first}"


class ReLU(_Leaf):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace


class MaxPool2d(_Leaf):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride or kernel_size, padding


class AdaptiveAvgPool2d(_Leaf):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size


class Dropout(_Leaf):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p


class BasicBlock(nn.Module):
    """torchvision BasicBlock child names: conv1 bn1 relu conv2 bn2 downsample."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        return _Leaf.forward(self, x)


class ResNet18(nn.Module):
    """torchvision ResNet-18 child names: conv1 bn1 relu maxpool layer1..4 avgpool fc.
    `fc` exists for state_dict parity and is never used by the hot path."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, planes in enumerate((64, 128, 256, 512), start=1):
            stride = 1 if i == 1 else 2
            down = None
            if stride != 1 or inplanes != planes:
                down = nn.Sequential(Conv2d(inplanes, planes, 1, stride, 0, bias=False), BatchNorm2d(planes))
            setattr(self, f"layer{i}", nn.Sequential(BasicBlock(inplanes, planes, stride, down),
                                                     BasicBlock(planes, planes)))
            inplanes = planes
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.fc = Linear(512, num_classes)

    def forward(self, x):
        return _Leaf.forward(self, x)


_warned_no_weights = False


def load_pretrained_resnet18(base_cnn, pretrained=True, frozen=False):
    """The reference always starts from ImageNet weights: `resnet18(weights=ResNet18_Weights.IMAGENET1K_V1)` downloads
    resnet18-f37072fd.pth by URL (Quadtree_from scratch/models.py:221, resnet/models.py:12,76, cnn+lstm/models.py:22).
    Nothing is fetched here.  QTCNN_RESNET18_WEIGHTS = path of a local torchvision ResNet-18 state_dict: it is loaded.
    Unset: the backbone keeps its random initialisation and a one-time UserWarning says so -- training heads on a
    frozen random backbone (resnet/, cnn+lstm/ variants) silently diverges from the reference otherwise.  Opt out
    with `pretrained=False` or QTCNN_RESNET18_WEIGHTS=none (tests, benchmarks, or when a full checkpoint is loaded
    with load_state_dict right after construction).  Returns True when weights were loaded."""
    global _warned_no_weights
    path = os.environ.get("QTCNN_RESNET18_WEIGHTS")
    if not pretrained or (path is not None and path.strip().lower() in ("none", "random", "0", "")):
        return False
    if path:
        base_cnn.load_state_dict(torch.load(path, map_location="cpu"))
        return True
    if not _warned_no_weights:
        _warned_no_weights = True
        import warnings
        warnings.warn(
            "QuadtreeCNN build: the reference initialises its ResNet-18 from ImageNet weights (downloaded by URL), "
            "which are not available offline; the backbone" + (" is FROZEN and" if frozen else "") + " keeps a RANDOM "
            "initialisation. Set QTCNN_RESNET18_WEIGHTS=/path/to/resnet18-f37072fd.pth (torchvision state_dict), load a "
            "checkpoint with load_state_dict(), or silence this with pretrained=False / QTCNN_RESNET18_WEIGHTS=none.",
            UserWarning, stacklevel=3)
    return False
