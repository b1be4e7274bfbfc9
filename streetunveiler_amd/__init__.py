"""streetunveiler_amd -- MI355X-native 2D-Gaussian (surfel) splatting rasterizer.

Only what the hot path needs: csrc/ (HIP kernels + C-ABI), the ctypes loader,
the host-side mirror of the reference operator surface, a synthetic scene
generator for the benchmark, and the frame-sharded multi-GPU helper.
"""
__version__ = "0.1.0"
