"""Drop-in for /root/reference/3dcnn/models.py (`from models import get_model`).

Quadtree3DCNN (:96-214) with the reference's constructor and `get_model` signature (:493-522); the Conv3d stack runs on
the MFMA kernels over time-major clips, one launch per 3x3x3 convolution and direction (27-tap implicit GEMM; block 1 from the
f32 clip, block 2 slab-resident: <pkg>/video3d.py).
The 2-D models this file also carries in the reference (StandardResNetCNN :34-92, QuadtreeCNN :381-490) are the ones of
the resnet/ drop-in.  ResNet3DVideo / HybridQuadtree3DCNN (:220-375) wrap torchvision's r3d_18 with KINETICS400 weights
(a remote fetch, SURVEY.md section 2 row 11: out of scope): they raise NotImplementedError.
"""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.basename(_PKG_DIR)
if os.path.dirname(_PKG_DIR) not in sys.path:
    sys.path.insert(0, os.path.dirname(_PKG_DIR))
_v3d = importlib.import_module(_PKG + ".video3d")
_impl = importlib.import_module(_PKG + ".quadtree")
QtError = importlib.import_module(_PKG + "._lib").QtError

StandardResNetCNN = _impl.StandardResNetCNN

import torch as _torch
import torch.nn.functional as _F


class FocalLoss(_torch.nn.Module):
    """Counterpart of /root/reference/3dcnn/models.py:8-47 (a public name of that file; its trainer does not use it).
    A torch-level loss like CrossEntropyLoss -- it sits above the hot path and stays PyTorch:
        loss_i = -alpha[t_i] * (1 - p_i)^gamma * log p_i,   p_i = softmax(logits_i)[t_i]
    `alpha`: a list of per-class weights (len == num_classes), or a float a, stored as [a, 1 - a] exactly as the reference
    does (usable for two classes only).  The reference leaves `alpha_t` unbound for every other combination and dies with
    UnboundLocalError; here that case is a ValueError naming the cause."""

    def __init__(self, alpha=0.25, gamma=2.0, reduction='mean', num_classes=None):
        super().__init__()
        self.gamma = gamma
        self.reduction = reduction
        self.num_classes = num_classes
        if isinstance(alpha, (float, int)):
            alpha = _torch.tensor([alpha, 1 - alpha])
        elif isinstance(alpha, list):
            alpha = _torch.tensor(alpha)
        self.alpha = alpha

    def forward(self, inputs, targets):
        logp = _F.log_softmax(inputs, dim=-1).gather(1, targets.view(-1, 1)).squeeze()
        p = logp.exp()
        a = self.alpha
        if _torch.is_tensor(a) and a.dim() > 1:
            alpha_t = a[targets]
        elif _torch.is_tensor(a) and a.dim() == 1 and a.size(0) == self.num_classes:
            alpha_t = a.to(targets.device)[targets]
        else:
            raise ValueError("FocalLoss: alpha must hold one weight per class (len == num_classes); the reference leaves "
                             "alpha_t undefined for this configuration")
        loss = -alpha_t * (1 - p).pow(self.gamma) * logp
        if self.reduction == 'mean':
            return loss.mean()
        if self.reduction == 'sum':
            return loss.sum()
        return loss


class Quadtree3DCNN(_v3d.Quadtree3DCNN):
    def __init__(self, num_classes, sequence_length=8, cnn_3d_feature_dim=1024, numerical_feature_dim=47, dropout_rate=0.6,
                 mode='quadtree_3d_fusion', **kw):
        super().__init__(num_classes, sequence_length, cnn_3d_feature_dim, numerical_feature_dim, dropout_rate, mode, **kw)


class QuadtreeCNN(_impl.QuadtreeCNN):
    def __init__(self, num_classes, cnn_feature_dim=512, numerical_feature_dim=47, dropout_rate=0.5, mode='fusion', **kw):
        super().__init__(num_classes, cnn_feature_dim, numerical_feature_dim, dropout_rate, mode=mode, freeze_backbone=True, **kw)


class ResNet3DVideo:
    def __init__(self, *a, **k):
        raise NotImplementedError("ResNet3DVideo wraps torchvision r3d_18 (KINETICS400 weights, a remote fetch): out of scope")


class HybridQuadtree3DCNN:
    def __init__(self, *a, **k):
        raise NotImplementedError("HybridQuadtree3DCNN wraps torchvision r3d_18 (KINETICS400 weights, a remote fetch): out of scope")


def get_model(num_classes, device, numerical_feature_dim=47, mode='fusion', sequence_length=8, print_num_params=True):
    if mode == 'standard_resnet_only':
        model = StandardResNetCNN(num_classes=num_classes).to(device)
    elif mode in ['quadtree_3d_fusion', 'quadtree_3d_image_only']:
        model = Quadtree3DCNN(num_classes=num_classes, sequence_length=sequence_length,
                              numerical_feature_dim=numerical_feature_dim, mode=mode, cnn_3d_feature_dim=1024).to(device)
    elif mode == 'resnet_3d_video_only':
        model = ResNet3DVideo(num_classes=num_classes)
    elif mode in ['hybrid_quadtree_3d_fusion', 'hybrid_quadtree_3d_image_only']:
        model = HybridQuadtree3DCNN(num_classes=num_classes, sequence_length=sequence_length,
                                    numerical_feature_dim=numerical_feature_dim, mode=mode)
    else:
        model = QuadtreeCNN(num_classes=num_classes, numerical_feature_dim=numerical_feature_dim, mode=mode).to(device)
    if print_num_params:
        num_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f"Number of trainable parameters: {num_params / 1e6:.2f} Million (Mode: {mode})")
    return model
