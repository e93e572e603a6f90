"""popcorn_amd -- MI355X-native implementation of POPCORN's dense per-pixel CNN path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the path
runs in hand-written HIP kernels for gfx950 (popcorn_amd/csrc -> libpopcorn_hip.so, C ABI in include/popcorn_hip.h).
"""
__version__ = "0.1.0"
