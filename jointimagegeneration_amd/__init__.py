"""guidegen-mi355x: MI355X-native sampling engine for GuideGen's two denoising hot paths.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all arithmetic on the hot path runs in hand-written HIP kernels for gfx950 behind
the C-ABI declared in include/guidegen_hip.h (libguidegen_hip.so).
"""
__version__ = "0.1.0"
