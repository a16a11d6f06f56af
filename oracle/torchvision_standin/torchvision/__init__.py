"""Offline stand-in for the `torchvision` package (TEST INFRASTRUCTURE ONLY).

torchvision is not installed in the build container and cannot be fetched.
The reference does `import torchvision.models as models`
(/root/reference/Quadtree_from scratch/models.py:4, resnet/models.py:4) only
to obtain the ResNet-18 module topology, so this package restates that
topology (see models/__init__.py) and nothing else.  It is put on sys.path by
tests/golden/make_golden.py when the reference is imported by path to produce
golden vectors; the product never imports it.
"""
from . import models  # noqa: F401
