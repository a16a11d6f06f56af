"""MI355X-native QuadtreeCNN hot path (forward/backward of the multimodal
Sun-Salutation pose classifier) behind the reference's nn.Module surface.

The directory name is not a Python identifier; import it with
    importlib.import_module("multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd")
or put `<this dir>/quadtree_from_scratch` (or `/resnet`) on sys.path and keep the
reference's `from models import get_model`.
"""
from ._lib import LIB_PATH, QtError  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .quadtree import AttentionHierarchicalCNN, CnnLstm, QuadtreeCNN, StandardResNetCNN  # noqa: F401
from .video3d import Ji3DCNN, Quadtree3DCNN  # noqa: F401

__all__ = ["QuadtreeCNN", "StandardResNetCNN", "AttentionHierarchicalCNN", "CnnLstm", "Quadtree3DCNN", "Ji3DCNN", "FusedAdam",
           "QtError", "LIB_PATH"]
