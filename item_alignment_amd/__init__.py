"""item_alignment_amd — MI355X-native engine for the item-pair matching train step.

Layout: csrc/ (HIP kernels + C ABI, built into libitemalign_hip.so), _lib.py (ctypes binding),
ops.py (raw op entry points), models/ (host-side mirror of the reference's src/models operator API).
"""
__all__ = ["_lib", "ops"]
