"""Drop-in for /root/reference/resnet/models.py (`from models import get_model`).

StandardResNetCNN (:7-65), QuadtreeCNN with `mode` and a frozen backbone (:70-180)
and get_model (:183-194) with the reference's signatures; arithmetic in gfx950 kernels.
"""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.basename(_PKG_DIR)
if os.path.dirname(_PKG_DIR) not in sys.path:
    sys.path.insert(0, os.path.dirname(_PKG_DIR))
_impl = importlib.import_module(_PKG + ".quadtree")
QtError = importlib.import_module(_PKG + "._lib").QtError
FusedAdam = importlib.import_module(_PKG + ".optim").FusedAdam  # optional replacement of optim.Adam(...)

StandardResNetCNN = _impl.StandardResNetCNN


class QuadtreeCNN(_impl.QuadtreeCNN):
    def __init__(self, num_classes, cnn_feature_dim=512, numerical_feature_dim=47, dropout_rate=0.5,
                 mode='fusion', **kw):
        super().__init__(num_classes, cnn_feature_dim, numerical_feature_dim, dropout_rate,
                         mode=mode, freeze_backbone=True, **kw)


def get_model(num_classes, device, numerical_feature_dim=47, mode='fusion', print_num_params=True):
    if mode == 'standard_resnet_only':
        model = StandardResNetCNN(num_classes=num_classes).to(device)
    else:
        model = QuadtreeCNN(num_classes=num_classes, numerical_feature_dim=numerical_feature_dim, mode=mode).to(device)
    if print_num_params:
        num_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f"Number of trainable parameters: {num_params / 1e6:.2f} Million (Mode: {mode})")
    return model
