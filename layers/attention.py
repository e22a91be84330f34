"""Drop-in module path of the reference (`layers/attention.py`): re-exports the MI355X build."""
from mmbidaf_amd.attention import BiDAFAttention, MultimodalAttentionDecoder, masked_softmax  # noqa: F401

__all__ = ["BiDAFAttention", "MultimodalAttentionDecoder", "masked_softmax"]
