"""Opt-in, zero-edit drop-in: make `import nerf_qa.<hot-path module>` resolve to this package.

The reference's scripts name the hot path by module path -- `from nerf_qa.DISTS_pytorch.DISTS_pt_original import
DISTS, prepare_image` (run_nerf_qa.py:29), `from nerf_qa.model_stats import NeRFQAModel` (run_nerf_qa.py:33),
`from nerf_qa.DISTS_pytorch.DISTS_pt import DISTS, prepare_image` (nerf_qa/data.py:34, prep.py:27),
`from nerf_qa.ADISTS import ADISTS` (prep.py:28) -- and they save / load WHOLE modules with pickle
(`torch.save(model, ...)` run_nerf_qa.py:502, `torch.load` reeval.py:83), which records those module paths.

    import nerf_qa_amd; nerf_qa_amd.install_alias()      # first lines of the entry script, or sitecustomize

registers the HIP-backed modules under the reference's names in `sys.modules`, so the scripts' import lines
and their pickles work unmodified.  If the reference checkout is importable, `nerf_qa` stays the real package
(its data loaders, NR models ... keep working) and only the hot-path submodules are replaced; if it is not, a
bare `nerf_qa` namespace is created that holds just these.  Nothing of the reference is copied or executed.
"""
from __future__ import annotations

import importlib
import sys
import types

# reference module path -> module of this package
ALIASES = {
    "nerf_qa.DISTS_pytorch": "nerf_qa_amd.DISTS_pytorch",
    "nerf_qa.DISTS_pytorch.DISTS_pt": "nerf_qa_amd.DISTS_pytorch.DISTS_pt",
    "nerf_qa.DISTS_pytorch.DISTS_pt_original": "nerf_qa_amd.DISTS_pytorch.DISTS_pt_original",
    "nerf_qa.DISTS_pytorch.DISTS_pt_softmax": "nerf_qa_amd.DISTS_pytorch.DISTS_pt_softmax",
    "nerf_qa.ADISTS": "nerf_qa_amd.ADISTS",
    "nerf_qa.ADISTS.ADISTS": "nerf_qa_amd.ADISTS.ADISTS",
    "nerf_qa.model": "nerf_qa_amd.model",
    "nerf_qa.model_stats": "nerf_qa_amd.model_stats",
}


def install_alias(force: bool = True) -> dict:
    """Register the aliases; returns {reference name: module}.  force=False keeps a hot-path module of the real
    package that is already imported (and reports it in the result under its own name)."""
    try:
        root = importlib.import_module("nerf_qa")  # the reference checkout, if it is on sys.path
    except ImportError:
        root = types.ModuleType("nerf_qa")
        root.__path__ = []  # a package with no files of its own
        root.__doc__ = "namespace created by nerf_qa_amd.install_alias(): only the DISTS / A-DISTS hot path"
        sys.modules["nerf_qa"] = root
    out = {}
    for ref_name in sorted(ALIASES, key=lambda n: n.count(".")):  # parents first
        if not force and ref_name in sys.modules:
            out[ref_name] = sys.modules[ref_name]
            continue
        mod = importlib.import_module(ALIASES[ref_name])
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent == "nerf_qa":  # deeper parents are this package's own modules: their attributes are already right
            setattr(root, leaf, mod)  # (nerf_qa_amd.ADISTS.ADISTS is the CLASS, re-exported as in the reference)
        out[ref_name] = mod
    return out


def remove_alias() -> None:
    """Undo install_alias() (tests)."""
    for ref_name in ALIASES:
        mod = sys.modules.get(ref_name)
        if mod is not None and mod.__name__.startswith("nerf_qa_amd"):
            del sys.modules[ref_name]
    root = sys.modules.get("nerf_qa")
    if root is not None and getattr(root, "__path__", None) == []:
        del sys.modules["nerf_qa"]
