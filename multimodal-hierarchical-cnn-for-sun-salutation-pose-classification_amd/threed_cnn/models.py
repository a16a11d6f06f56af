"""Drop-in for /root/reference/3dcnn/models.py (`from models import get_model`).

Quadtree3DCNN (:96-214) with the reference's constructor and `get_model` signature (:493-522); the Conv3d stack runs on
the MFMA implicit-GEMM kernels (three 2-D launches per 3x3x3 convolution over time-major clips, <pkg>/video3d.py).
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
