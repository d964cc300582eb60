"""msspe_amd -- Python host layer over libmsspe_hip.so (ctypes; no compute happens in Python).

The product path is the HIP library; importing this package never touches oracle/.
"""
from .capi import (Engine, Group, group_rows, MsspeError, Chem, KmerOpt, lib_path, load_library, pack_oligos,
                   unpack_oligo, round_g_f32, round_fixed_f32, g_cut, STATUS)
from . import synth
from . import distributed

__all__ = ["Engine", "Group", "group_rows", "MsspeError", "Chem", "KmerOpt", "lib_path", "load_library", "pack_oligos",
           "unpack_oligo", "round_g_f32", "round_fixed_f32", "g_cut", "STATUS", "synth", "distributed"]
