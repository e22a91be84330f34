"""Drop-in module path of the reference (`models.py`): `from models import MMBiDAF` (train.py:23)."""
from mmbidaf_amd.model import MMBiDAF  # noqa: F401

__all__ = ["MMBiDAF"]
