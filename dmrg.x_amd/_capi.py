"""ctypes binding of include/dmrgx.h (libdmrgx_hip.so).  No compute happens in Python.

The library is the product: if it is missing or cannot be loaded this module raises -- there is no CPU
fallback (the CPU restatement under oracle/ is test infrastructure and is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMRGX_LIB: another build of the SAME library (tools/ab.sh bench: same-box A/B of saved builds) -- never a different implementation; the
# product file itself is never overwritten by a tool
LIB_PATH = os.environ.get("DMRGX_LIB") or os.path.join(_HERE, "libdmrgx_hip.so")

DMRGX_OK = 0
DMRGX_ERR_ARG = 62
DMRGX_ERR_OUTOFRANGE = 63
DMRGX_ERR_DEVICE = 97
DMRGX_ERR_NOTCONV = 91
CELL_DENSE, CELL_IDENT = 1, 2


class DmrgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dmrgx status {code}: {msg}")
        self.code = code


class Cell(C.Structure):
    _fields_ = [("row_sector", C.c_int32), ("r0", C.c_int32), ("c0", C.c_int32), ("nr", C.c_int32), ("nc", C.c_int32),
                ("kind", C.c_int32), ("scale", C.c_double), ("data", C.c_void_p), ("ld", C.c_int64)]


class SecOp(C.Structure):
    _fields_ = [("shift", C.c_int32), ("transposed", C.c_int32), ("ncells", C.c_int32), ("cells", C.POINTER(Cell))]


class Sectors(C.Structure):
    _fields_ = [("nsec", C.c_int32), ("size", C.POINTER(C.c_int32))]


class Term(C.Structure):
    _fields_ = [("a", C.c_double), ("left_op", C.c_int32), ("right_op", C.c_int32)]


class KronDesc(C.Structure):
    _fields_ = [("left", Sectors), ("right", Sectors), ("nblocks", C.c_int32),
                ("block_il", C.POINTER(C.c_int32)), ("block_ir", C.POINTER(C.c_int32)),
                ("n_left_ops", C.c_int32), ("n_right_ops", C.c_int32),
                ("left_ops", C.POINTER(SecOp)), ("right_ops", C.POINTER(SecOp)),
                ("h_left", C.POINTER(SecOp)), ("h_right", C.POINTER(SecOp)),
                ("nterms", C.c_int32), ("terms", C.POINTER(Term)),
                ("world_size", C.c_int32), ("rank", C.c_int32)]


class RdmReport(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_sweeps", "solver", "trid_persistent_matrices", "trid_launch_matrices", "max_workgroups_per_matrix",
                                         "merge_levels", "wy_blocks_max", "timed_out", "process_timeouts", "persistent_off")]


class RdmVecTask(C.Structure):
    _fields_ = [("side", C.c_int32), ("k", C.c_int32), ("count", C.c_int32), ("pad", C.c_int32), ("dst_dev", C.c_void_p), ("ld", C.c_int64)]


class KronInfo(C.Structure):
    _fields_ = [("n_states", C.c_int64), ("vec_len", C.c_int64), ("local_offset", C.c_int64), ("local_len", C.c_int64),
                ("seg_stride", C.c_int64), ("flops_alg", C.c_double), ("bytes_alg", C.c_double), ("flops_exec", C.c_double),
                ("bytes_workspace", C.c_double), ("n_groups", C.c_int32), ("n_tiles_stage1", C.c_int32), ("n_tiles_stage2", C.c_int32),
                ("n_tiles_big", C.c_int32), ("flops_alg_big", C.c_double)]


class AxpyTask(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("dst_base", C.c_void_p), ("src", C.c_void_p), ("ldd", C.c_int64), ("lds", C.c_int64),
                ("nr", C.c_int32), ("nc", C.c_int32), ("transposed", C.c_int32), ("alpha", C.c_double)]


class Dot2dTask(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int64), ("b", C.c_void_p), ("ldb", C.c_int64), ("nr", C.c_int32), ("nc", C.c_int32), ("out", C.c_int32), ("pad", C.c_int32)]


class GemmTask(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("accumulate", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64)]


class Rotation(C.Structure):
    _fields_ = [("n_new", C.c_int32), ("old_sector", C.POINTER(C.c_int32)), ("kept", C.POINTER(C.c_int32)), ("rot_t", C.POINTER(C.c_void_p))]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class EigsOpts(C.Structure):
    _fields_ = [("ncv", C.c_int32), ("max_it", C.c_int32), ("tol", C.c_double), ("seed", C.c_uint64),
                ("use_initial", C.c_int32), ("max_matvec", C.c_int32), ("allgather", ALLGATHER_FN), ("allreduce_sum", ALLREDUCE_FN), ("user", C.c_void_p),
                ("comm", C.c_void_p), ("method", C.c_int32), ("min_initial_norm2", C.c_double), ("gd_minv", C.c_int32), ("reserved_", C.c_int32)]


class EigsStats(C.Structure):
    _fields_ = [("n_matvec", C.c_int32), ("n_restart", C.c_int32), ("converged", C.c_int32), ("start_rejected", C.c_int32),
                ("residual", C.c_double), ("seconds", C.c_double)]


# name -> (restype, argtypes): every symbol include/dmrgx.h declares
SIGNATURES = {
    "dmrgx_abi_version": (C.c_int32, []),
    "dmrgx_last_error": (C.c_char_p, []),
    "dmrgx_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "dmrgx_kron_plan_create": (C.c_int32, [C.POINTER(KronDesc), C.c_void_p, C.POINTER(C.c_void_p)]),
    "dmrgx_kron_plan_info": (C.c_int32, [C.c_void_p, C.POINTER(KronInfo)]),
    "dmrgx_kron_apply": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmrgx_kron_plan_destroy": (C.c_int32, [C.c_void_p]),
    "dmrgx_kron_diag": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmrgx_kron_plan_timing": (C.c_int32, [C.c_void_p, C.c_int32]),
    "dmrgx_kron_plan_timing_read": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dmrgx_kron_vec_to_striped": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmrgx_kron_vec_from_striped": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmrgx_stripe_bounds": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dmrgx_stripe_bounds_of_block": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dmrgx_dgemm_nn": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_int64, C.c_void_p]),
    "dmrgx_rdm_create": (C.c_int32, [C.POINTER(Sectors), C.POINTER(Sectors), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dmrgx_rdm_eigenvalues": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "dmrgx_rdm_create_warm": (C.c_int32, [C.POINTER(Sectors), C.POINTER(Sectors), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                               C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dmrgx_rdm_eigenvectors": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "dmrgx_rdm_info": (C.c_int32, [C.c_void_p, C.POINTER(RdmReport)]),
    "dmrgx_rdm_select": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "dmrgx_rdm_eigenvectors_batch": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(RdmVecTask), C.c_void_p]),
    "dmrgx_rdm_destroy": (C.c_int32, [C.c_void_p]),
    "dmrgx_cells_axpy": (C.c_int32, [C.c_int32, C.POINTER(AxpyTask), C.c_void_p]),
    "dmrgx_rotate_ops": (C.c_int32, [C.POINTER(Sectors), C.POINTER(Rotation), C.c_int32, C.POINTER(SecOp), C.POINTER(C.POINTER(C.c_void_p)), C.c_void_p]),
    "dmrgx_malloc": (C.c_int32, [C.POINTER(C.c_void_p), C.c_size_t]),
    "dmrgx_free": (C.c_int32, [C.c_void_p]),
    "dmrgx_mem_stats": (C.c_int32, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "dmrgx_memcpy_h2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmrgx_memcpy_d2h": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmrgx_memcpy_d2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmrgx_memset_zero": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmrgx_stream_sync": (C.c_int32, [C.c_void_p]),
    "dmrgx_dgemm_batch": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p]),
    "dmrgx_dot": (C.c_int32, [C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "dmrgx_dot_async": (C.c_int32, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmrgx_dot2d_batch": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmrgx_eigs_comm_timing": (C.c_int32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "dmrgx_eigs_lowest": (C.c_int32, [C.c_void_p, C.POINTER(EigsOpts), C.POINTER(C.c_double), C.c_void_p,
                                      C.POINTER(EigsStats), C.c_void_p]),
    "dmrgx_rdm_create_subset": (C.c_int32, [C.POINTER(Sectors), C.POINTER(Sectors), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dmrgx_set_device": (C.c_int32, [C.c_int32]),
    "dmrgx_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "dmrgx_comm_init": (C.c_int32, [C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dmrgx_comm_init_host_staged": (C.c_int32, [C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)]),
    "dmrgx_comm_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dmrgx_comm_allgather": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dmrgx_comm_allreduce_sum": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dmrgx_comm_bcast": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "dmrgx_comm_allgather_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmrgx_comm_barrier": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "dmrgx_comm_destroy": (C.c_int32, [C.c_void_p]),
}

_lib = None


def lib():
    """Load libdmrgx_hip.so (built by `make` / __graft_entry__.build()).  Raises if absent: no fallback."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime (same SONAME as /opt/rocm's).  It must be the copy the process loads first,
        # otherwise torch later finds "No HIP GPUs": import torch before dlopen-ing the kernel library.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make` (hipcc --offload-arch=gfx950). "
                              "The dmrgx hot path has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI symbol is missing
            fn.restype, fn.argtypes = res, args
        if L.dmrgx_abi_version() != 3:
            raise ImportError("dmrgx ABI version mismatch")
        _lib = L
    return _lib


def check(status):
    if status != DMRGX_OK:
        raise DmrgxError(status, lib().dmrgx_last_error().decode(errors="replace"))


def require_device():
    n = C.c_int32(0)
    check(lib().dmrgx_device_count(C.byref(n)))
    return n.value
