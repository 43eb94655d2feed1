"""elf_amd -- MI355X-native batched Go environment behind ELF's DarkForest Go batch API.

Only what the hot path needs lives here: the HIP kernels + C-ABI (csrc/, libgoenv.so),
the ctypes binding (goenv.py) and the host-side mirror of the reference's
GameContext / GCWrapper interface (context.py, utils_elf.py).
"""
__all__ = ["goenv"]
