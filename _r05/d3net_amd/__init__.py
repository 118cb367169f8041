"""d3net_amd -- MI355X (gfx950) implementation of D3Net's PointGroup hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd glue); all compute on the
path runs in hand-written HIP kernels behind the C ABI of include/d3hip.h (libd3hip.so).
There is no CPU fallback: importing the operator modules without the library, or calling them
on CPU tensors, raises.
"""
__version__ = "0.1.0"
