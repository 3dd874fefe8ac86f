"""muchsalsa_amd -- MI355X-native overlap core for MuCHSALSA (host-side Python view of include/msgpu.h).

The compute lives in libmsgpu.so (hand-written HIP for gfx950); nothing here computes on the CPU.
"""
__all__ = ["overlap", "sequences", "distributed", "synth"]
