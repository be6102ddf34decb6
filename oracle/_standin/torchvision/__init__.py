"""Authoring-container stand-in for the absent `torchvision` package.

TEST TOOLING.  It exists only so that oracle/make_goldens.py can import the
reference's own nerf_qa.DISTS_pytorch / nerf_qa.ADISTS modules from
/root/reference (they do `from torchvision import models, transforms` at import
time) and run them on CPU to pin the oracle.  It never travels into the product
path and is not on sys.path anywhere else.
"""
from . import models, transforms  # noqa: F401
