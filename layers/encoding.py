"""Drop-in module path of the reference (`layers/encoding.py`): re-exports the MI355X build."""
from mmbidaf_amd.encoding import Embedding, HighwayEncoder, ImageEmbedding, RNNEncoder  # noqa: F401

__all__ = ["Embedding", "HighwayEncoder", "ImageEmbedding", "RNNEncoder"]
