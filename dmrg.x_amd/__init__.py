"""dmrgx_amd -- MI355X-native hot path of the DMRG.x sweep engine.

Layout:
  csrc/          HIP kernels for gfx950 + the C ABI (include/dmrgx.h)  -> libdmrgx_hip.so
  _capi.py       ctypes binding of the C ABI
  superblock.py  host mirror of the reference's shell-matrix interface (KronSumConstruct / MatMult / destroy / EPS)
  workloads.py   synthetic superblocks with the sector structure of BASELINE.json's configs

The directory name contains a dot, so it is loaded through __graft_entry__.load_package() under the module
name ``dmrgx_amd``.  Importing it never touches oracle/ and never falls back to a CPU path.
"""
from . import _capi, workloads  # noqa: F401
