from .DISTS_pt import DISTS  # same re-export as nerf_qa/DISTS_pytorch/__init__.py:1
