"""rad_amd — MI355X (gfx950) implementation of RAD's HNSW neighbor-expansion hot path.

Host side is plain Python over a thin C-ABI HIP library (rad_amd/csrc ->
rad_amd/_build/librad_hip.so, declared in include/rad_hip.h).  There is no CPU
fallback and no PyTorch dependency in this package.
"""
__version__ = "0.1.0"
