"""MI355X-native DISTS / A-DISTS hot path of kobejean/nerf-qa."""
from ._lib import NqaError  # noqa: F401
from .alias import install_alias, remove_alias  # noqa: F401

__all__ = ["NqaError", "install_alias", "remove_alias"]
