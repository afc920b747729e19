"""MI355X-native blind-bid Bulletproofs engine: thin ctypes binding over libbbp_hip.so (include/bbp.h).

The library has no CPU compute path: importing works anywhere (so the C-ABI surface can be inspected), but
`Context()` raises unless a gfx950 device is present, and the import itself raises if the HIP library was not built.
"""
from ._native import (Context, Pool, BbpError, lib, lib_path, STATUS, SIGNATURES, record_size, entropy_size,
                      LAYOUT_BLIND_G_H, LAYOUT_BLIND_G, BASE_BBLIND, BASE_G0, BASE_H0, BASE_B, NUM_BASES, STREAM_CONTEXT,
                      compile_circuit)

__all__ = ["Context", "Pool", "BbpError", "lib", "lib_path", "STATUS", "SIGNATURES", "record_size", "entropy_size",
           "LAYOUT_BLIND_G_H", "LAYOUT_BLIND_G", "BASE_BBLIND", "BASE_G0", "BASE_H0", "BASE_B", "NUM_BASES", "STREAM_CONTEXT", "compile_circuit"]
