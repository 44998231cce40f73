"""ctypes wrapper of oracle/kron_ref.c (literal MatMult_KronSumShell).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_kron.so")
_lib = None


class _Ctx(C.Structure):
    _fields_ = [("N", C.c_int64), ("nterms", C.c_int32), ("nblocks", C.c_int32), ("term_a", C.c_void_p),
                ("A_indptr", C.c_void_p), ("A_indices", C.c_void_p), ("A_data", C.c_void_p),
                ("B_indptr", C.c_void_p), ("B_indices", C.c_void_p), ("B_data", C.c_void_p),
                ("Rows_L", C.c_void_p), ("Rows_R", C.c_void_p), ("blk_of_row", C.c_void_p),
                ("bks_L", C.c_void_p), ("col_NStatesR", C.c_void_p), ("fws_O", C.c_void_p), ("valid", C.c_void_p)]


def build():
    """Compile the C restatement (gcc, OpenMP).  Called by __graft_entry__.build() and lazily by tests."""
    src = os.path.join(_HERE, "kron_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", src, "-o", _SO])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_kron_apply_ref.restype = C.c_int
        _lib.oracle_kron_apply_ref.argtypes = [C.POINTER(_Ctx), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        _lib.oracle_kron_ref_flops.restype = C.c_int64
        _lib.oracle_kron_ref_flops.argtypes = [C.POINTER(_Ctx), C.c_int64, C.c_int64]
    return _lib


class ShellApplyC:
    """Literal row loop of src/DMRGKron.cpp:1842-1864 over the descriptors of oracle.kron.ShellCtx."""

    def __init__(self, shell):
        self.shell = shell
        self._keep = []

        def i64(a):
            a = np.ascontiguousarray(a, dtype=np.int64)
            self._keep.append(a)
            return a

        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._keep.append(a)
            return a

        def ptr_array(arrs):
            p = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
            self._keep.append(p)
            return C.cast(p, C.c_void_p)

        nt = shell.nterms
        Ai = [i64(m.indptr) for m in shell.A]; Aj = [i64(m.indices) for m in shell.A]; Av = [f64(m.data) for m in shell.A]
        Bi = [i64(m.indptr) for m in shell.B]; Bj = [i64(m.indices) for m in shell.B]; Bv = [f64(m.data) for m in shell.B]
        c = _Ctx()
        c.N, c.nterms, c.nblocks = shell.N, nt, shell.kb.size()
        c.term_a = f64(shell.term_a).ctypes.data
        c.A_indptr, c.A_indices, c.A_data = ptr_array(Ai), ptr_array(Aj), ptr_array(Av)
        c.B_indptr, c.B_indices, c.B_data = ptr_array(Bi), ptr_array(Bj), ptr_array(Bv)
        c.Rows_L, c.Rows_R = i64(shell.Rows_L).ctypes.data, i64(shell.Rows_R).ctypes.data
        c.blk_of_row = i64(shell.blk_of_row).ctypes.data
        c.bks_L, c.col_NStatesR = i64(shell.bks_L).ctypes.data, i64(shell.col_NStatesR).ctypes.data
        c.fws_O, c.valid = i64(shell.fws_O).ctypes.data, i64(shell.valid).ctypes.data
        self.ctx = c
        self.threads_used = 0

    def apply(self, x, row_begin=0, row_end=None, nthreads=0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.shell.N)
        row_end = self.shell.N if row_end is None else row_end
        self.threads_used = lib().oracle_kron_apply_ref(C.byref(self.ctx), x.ctypes.data, y.ctypes.data, row_begin, row_end, nthreads)
        return y

    def flops(self, row_begin=0, row_end=None):
        row_end = self.shell.N if row_end is None else row_end
        return lib().oracle_kron_ref_flops(C.byref(self.ctx), row_begin, row_end)
