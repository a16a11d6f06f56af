"""Offline stand-in for `torchvision.models.video` (test infrastructure only).

/root/reference/3dcnn/models.py:5 imports this module at the top of the file; only its HybridQuadtree3DCNN /
ResNet3DVideo classes (out of scope: they need the KINETICS400 r3d_18 weights, a remote fetch) ever call it.  The
golden-vector generator needs the import to succeed to reach Quadtree3DCNN; nothing here is ever executed by a test.
"""


class R3D_18_Weights:
    KINETICS400_V1 = "KINETICS400_V1"
    DEFAULT = "KINETICS400_V1"


def r3d_18(weights=None, **kwargs):
    raise NotImplementedError("torchvision.models.video.r3d_18 is not available offline (out of scope: SURVEY.md section 2, row 11)")
