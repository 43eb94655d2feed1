"""elf_amd -- MI355X-native batched Go environment behind ELF's DarkForest Go batch API.

Only what the hot path needs lives here: the HIP kernels + C-ABI (csrc/, libgoenv.so), the
ctypes binding (goenv.py), the `go_game` pybind module that mirrors the reference's GameContext
(csrc/go_game/) and the Python caller's glue (elf_wrapper.py, go_loader.py, sharding.py).
"""
__all__ = ["goenv"]
