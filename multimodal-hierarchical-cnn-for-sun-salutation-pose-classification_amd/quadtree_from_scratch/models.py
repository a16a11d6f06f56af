"""Drop-in for /root/reference/Quadtree_from scratch/models.py (`from models import get_model`).

Same public names and signatures as the reference file (:214-325); the arithmetic
runs in hand-written gfx950 kernels.  All parameters are trainable in this
variant (the reference does not freeze the backbone here, :221).
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


class QuadtreeCNN(_impl.QuadtreeCNN):
    def __init__(self, num_classes, cnn_feature_dim=512, numerical_feature_dim=47, dropout_rate=0.5, **kw):
        super().__init__(num_classes, cnn_feature_dim, numerical_feature_dim, dropout_rate,
                         mode="fusion", freeze_backbone=False, **kw)


class HierarchicalQuadtreeCNN:
    """The reference class (:105-210) cannot run: its bottom-right slices use `w:` / `qw:`
    (empty, :182,196) and the conv raises "Kernel size can't be greater than actual input
    size".  The drop-in keeps that observable behaviour instead of inventing semantics."""

    def __init__(self, *a, **k):
        raise RuntimeError("HierarchicalQuadtreeCNN is unrunnable in the reference (empty slice at "
                           "models.py:182,196 -> RuntimeError in forward); not provided")


class AttentionHierarchicalCNN(_impl.AttentionHierarchicalCNN):
    """reference models.py:6-101: same constructor, state_dict keys and forward"""

    def __init__(self, num_classes, numerical_feature_dim=47, dropout_rate=0.5, **kw):
        super().__init__(num_classes, numerical_feature_dim, dropout_rate, **kw)


class StandardMultimodalCNN:
    """The reference defines this class as `pass` (:211-213); get_model's fall-through
    therefore raises TypeError for any other model name.  Same here."""
    pass


def get_model(model_name, num_classes, device, print_num_params=True):
    model_name = model_name.lower()
    if model_name == 'quadtree':
        model = QuadtreeCNN(num_classes=num_classes).to(device)
    elif model_name == 'hierarchical_quadtree':
        model = HierarchicalQuadtreeCNN(num_classes=num_classes)
    elif model_name == 'attention_hierarchical':
        model = AttentionHierarchicalCNN(num_classes=num_classes).to(device)
    else:
        model = StandardMultimodalCNN(backbone_name=model_name, num_classes=num_classes).to(device)
    if print_num_params:
        num_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f"Model: '{model_name.upper()}' | Trainable Parameters: {num_params / 1e6:.2f} Million")
    return model
