from .ADISTS import ADISTS  # same re-export as nerf_qa/ADISTS/__init__.py:1
