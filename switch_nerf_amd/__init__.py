"""switch_nerf_amd - MI355X (gfx950) native Switch-NeRF train hot path.

The compute lives in libswn_hip.so (hand-written HIP, C ABI declared in include/swn.h); this package is the
host-side mirror of the reference's Python interface (moe_layer / NeRFMoE / render_rays / training step).
There is no CPU fallback: importing `switch_nerf_amd.ops` without the built library raises.
"""
__version__ = "0.1.0"
