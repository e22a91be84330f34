"""mmbidaf_amd -- MI355X-native hot path of MMBiDAF (BiDAF attention + BiLSTM encoders).

Host side (Python, mirrors the reference's nn.Module API) over a C-ABI HIP library
(`libmmbidaf_hip.so`, include/mmbidaf.h).  PyTorch only provides device memory, streams,
autograd plumbing and torch.distributed.
"""
__version__ = "0.1.0"
