"""Host-side handle of the HIP superblock plan, mirroring the reference's shell-matrix life cycle:

    KronBlocks_t::KronSumConstruct(Terms, H)  ->  KronPlan(superblock)          (src/DMRGKron.cpp:759-841,1871-1917)
    MatMult(H, x, y)                          ->  KronPlan.apply(x, y)          (src/DMRGKron.cpp:1827-1869)
    MatDestroy_KronSumShell(&H)               ->  KronPlan.destroy()            (src/DMRGKron.cpp:1919-1942)
    EPSSolve(H) (EPS_HEP, SMALLEST_REAL)      ->  KronPlan.eigs_lowest()        (include/DMRGBlockContainer.hpp:1488-1499)

torch is used only to own device memory and streams; every computation goes through the C ABI (_capi).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .workloads import OpSm, OpSz, OpSp, CELL_DENSE, CELL_IDENT


def _i32(values):
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr


class KronPlan:
    def __init__(self, sb, device="cuda:0", world_size=1, rank=0, stream=None):
        _capi.require_device()          # fails loudly without a GPU: no CPU fallback
        self.sb = sb
        self.device = torch.device(device)
        self._keep = []                 # ctypes arrays + device tensors that must outlive plan creation
        L = _capi.lib()

        def secop(op, transposed=False, shift=None):
            cells = (_capi.Cell * max(len(op.cells), 1))()
            for i, c in enumerate(op.cells):
                cells[i].row_sector, cells[i].r0, cells[i].c0, cells[i].nr, cells[i].nc = c.row_sector, c.r0, c.c0, c.nr, c.nc
                cells[i].kind, cells[i].scale = c.kind, c.scale
                if c.kind == CELL_DENSE:
                    t = torch.from_numpy(np.ascontiguousarray(c.array, dtype=np.float64)).to(self.device)
                    self._keep.append(t)
                    cells[i].data, cells[i].ld = t.data_ptr(), c.nc
            self._keep.append(cells)
            s = _capi.SecOp()
            s.shift = op.shift if shift is None else shift
            s.transposed = 1 if transposed else 0
            s.ncells = len(op.cells)
            s.cells = cells
            return s

        # distinct (op, site) operators per side in term order; Sm(i) = transposed Sp(i) (never materialised)
        def side(ops, which):
            index, lst = {}, []
            for t in sb.terms:
                key = (t[1], t[2]) if which == 0 else (t[3], t[4])
                if key in index:
                    continue
                op, site = key
                if op == OpSm:
                    lst.append(secop(ops[(OpSp, site)], transposed=True, shift=-1))
                else:
                    lst.append(secop(ops[(op, site)]))
                index[key] = len(lst) - 1
            arr = (_capi.SecOp * max(len(lst), 1))(*lst)
            self._keep.append(arr)
            return index, arr, len(lst)

        li, larr, nl = side(sb.left_ops, 0)
        ri, rarr, nr = side(sb.right_ops, 1)
        terms = (_capi.Term * max(len(sb.terms), 1))()
        for i, t in enumerate(sb.terms):
            terms[i].a, terms[i].left_op, terms[i].right_op = t[0], li[(t[1], t[2])], ri[(t[3], t[4])]
        hl, hr = secop(sb.h_left), secop(sb.h_right)
        d = _capi.KronDesc()
        ls, rs = _i32(sb.left_sizes), _i32(sb.right_sizes)
        bil, bir = _i32([b[0] for b in sb.blocks]), _i32([b[1] for b in sb.blocks])
        d.left.nsec, d.left.size = len(sb.left_sizes), ls
        d.right.nsec, d.right.size = len(sb.right_sizes), rs
        d.nblocks, d.block_il, d.block_ir = len(sb.blocks), bil, bir
        d.n_left_ops, d.n_right_ops, d.left_ops, d.right_ops = nl, nr, larr, rarr
        d.h_left, d.h_right = C.pointer(hl), C.pointer(hr)
        d.nterms, d.terms = len(sb.terms), terms
        d.world_size, d.rank = world_size, rank
        self._handle = C.c_void_p()
        st = self._stream_ptr(stream)
        with torch.cuda.device(self.device):
            _capi.check(L.dmrgx_kron_plan_create(C.byref(d), st, C.byref(self._handle)))
        self._keep.clear()              # the plan owns copies of every operator
        info = _capi.KronInfo()
        _capi.check(L.dmrgx_kron_plan_info(self._handle, C.byref(info)))
        self.info = info
        self.world_size, self.rank = world_size, rank

    @staticmethod
    def _stream_ptr(stream):
        if stream is None:
            stream = torch.cuda.current_stream()
        return C.c_void_p(stream.cuda_stream)

    def new_vector(self):
        return torch.zeros(self.info.vec_len, dtype=torch.float64, device=self.device)

    def apply(self, x_full, y_local, stream=None):
        """y_local <- (H x_full)[this rank's segment]  ==  MatMult_KronSumShell."""
        assert x_full.dtype == torch.float64 and y_local.dtype == torch.float64
        assert x_full.numel() >= self.info.vec_len and y_local.numel() >= self.info.local_len
        _capi.check(_capi.lib().dmrgx_kron_apply(self._handle, C.c_void_p(x_full.data_ptr()), C.c_void_p(y_local.data_ptr()),
                                                 self._stream_ptr(stream)))

    def to_striped(self, v_ref, v_full, stream=None):
        _capi.check(_capi.lib().dmrgx_kron_vec_to_striped(self._handle, C.c_void_p(v_ref.data_ptr()), C.c_void_p(v_full.data_ptr()),
                                                          self._stream_ptr(stream)))

    def from_striped(self, v_full, v_ref, stream=None):
        _capi.check(_capi.lib().dmrgx_kron_vec_from_striped(self._handle, C.c_void_p(v_full.data_ptr()), C.c_void_p(v_ref.data_ptr()),
                                                            self._stream_ptr(stream)))

    def eigs_lowest(self, ncv=16, max_it=1000, tol=1e-8, seed=1, psi0=None, allgather=None, allreduce=None, stream=None,
                    max_matvec=0, comm=None, method=0, min_initial_norm2=0.0, gd_minv=0):
        """Lowest eigenpair (EPS_HEP / EPS_SMALLEST_REAL / nev=1).  Returns (e0, psi_full tensor, stats).

        max_matvec > 0 (benchmarks): run exactly that many Lanczos steps; non-convergence is then not an error.
        comm: a Communicator -- the solver then issues its RCCL collectives itself (no Python inside the solve); the
        allgather/allreduce callbacks are the harness alternative."""
        opts = _capi.EigsOpts()
        opts.ncv, opts.max_it, opts.tol, opts.seed, opts.max_matvec = ncv, max_it, tol, seed, max_matvec
        psi = self.new_vector()
        if psi0 is not None:
            psi.copy_(psi0)
            opts.use_initial = 1
        self._cb = (_capi.ALLGATHER_FN(allgather) if allgather else _capi.ALLGATHER_FN(),
                    _capi.ALLREDUCE_FN(allreduce) if allreduce else _capi.ALLREDUCE_FN())
        opts.allgather, opts.allreduce_sum = self._cb
        opts.comm = comm.handle if comm is not None else None
        opts.method = method            # 0: thick-restart Lanczos, 1: generalized Davidson (diagonal preconditioner)
        opts.gd_minv = gd_minv          # method 1: Ritz vectors kept at a restart beside the previous Ritz vector (0: default 1)
        opts.min_initial_norm2 = min_initial_norm2      # > 0: psi0 is dropped (stats.start_rejected) when |psi0|^2 is below this
        e0 = C.c_double(0.0)
        stats = _capi.EigsStats()
        rc = _capi.lib().dmrgx_eigs_lowest(self._handle, C.byref(opts), C.byref(e0), C.c_void_p(psi.data_ptr()),
                                           C.byref(stats), self._stream_ptr(stream))
        if not (rc == _capi.DMRGX_ERR_NOTCONV and max_matvec > 0):
            _capi.check(rc)
        return e0.value, psi, stats

    def timing(self, enable):
        _capi.check(_capi.lib().dmrgx_kron_plan_timing(self._handle, 1 if enable else 0))

    def timing_read(self):
        """-> ([ms stage-1 128-tiles, ms stage-1 64-tiles, ms stage-2 128-tiles, ms stage-2 64-tiles], applies recorded)."""
        ms, n = (C.c_double * 4)(), C.c_int64(0)
        _capi.check(_capi.lib().dmrgx_kron_plan_timing_read(self._handle, ms, C.byref(n)))
        return list(ms), n.value

    def destroy(self):
        if self._handle:
            _capi.check(_capi.lib().dmrgx_kron_plan_destroy(self._handle))
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def dgemm_nn(A, B, out=None):
    """C = A @ B through the MFMA grouped-GEMM kernel (row-major f64 device tensors)."""
    assert A.dtype == torch.float64 and B.dtype == torch.float64 and A.is_contiguous() and B.is_contiguous()
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    if out is None:
        out = torch.empty((M, N), dtype=torch.float64, device=A.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _capi.check(_capi.lib().dmrgx_dgemm_nn(M, N, K, C.c_void_p(A.data_ptr()), K, C.c_void_p(B.data_ptr()), N,
                                           C.c_void_p(out.data_ptr()), N, st))
    return out


class ReducedDensityMatrices:
    """Device RDM blocks + spectra of a superblock state (GetTruncation's rank-0 loop,
    include/DMRGBlockContainer.hpp:1715-1775).  psi: device tensor in the reference's vector layout."""

    def __init__(self, left_sizes, right_sizes, blocks, psi, warm=None):
        """warm: optional {(side, k): (n x n) device tensor of eigenvectors as rows from a previous solve} (warm start)."""
        L = _capi.lib()
        self.left_sizes, self.right_sizes, self.blocks = list(left_sizes), list(right_sizes), list(blocks)
        ls, rs = _i32(left_sizes), _i32(right_sizes)
        sl, sr = _capi.Sectors(len(left_sizes), ls), _capi.Sectors(len(right_sizes), rs)
        bil, bir = _i32([b[0] for b in blocks]), _i32([b[1] for b in blocks])
        self._handle = C.c_void_p()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if warm:
            ptrs = (C.c_void_p * (2 * len(blocks)))()
            for (side, k), t in warm.items():
                assert t.is_contiguous() and t.dtype == torch.float64 and t.shape == (self.size(side, k),) * 2
                ptrs[2 * k + side] = t.data_ptr()
            _capi.check(L.dmrgx_rdm_create_warm(C.byref(sl), C.byref(sr), len(blocks), bil, bir, C.c_void_p(psi.data_ptr()),
                                                C.cast(ptrs, C.c_void_p), st, C.byref(self._handle)))
        else:
            _capi.check(L.dmrgx_rdm_create(C.byref(sl), C.byref(sr), len(blocks), bil, bir, C.c_void_p(psi.data_ptr()), st, C.byref(self._handle)))
        self.report = _capi.RdmReport()         # which solver path ran: tridiagonalisation kind, workgroups per matrix, merge levels, time-outs
        _capi.check(L.dmrgx_rdm_info(self._handle, C.byref(self.report)))
        self.sweeps = self.report.n_sweeps

    def size(self, side, k):
        return (self.left_sizes[self.blocks[k][0]], self.right_sizes[self.blocks[k][1]])[side]

    def eigenvalues(self, side, k):
        out = (C.c_double * self.size(side, k))()
        _capi.check(_capi.lib().dmrgx_rdm_eigenvalues(self._handle, side, k, out))
        return np.array(out)

    def select(self, counts):
        """Second phase (dmrgx_rdm_select): form the eigenvectors of the counts[2*k + side] largest eigenvalues of every density matrix only."""
        arr = (C.c_int32 * (2 * len(self.blocks)))(*[int(c) for c in counts])
        _capi.check(_capi.lib().dmrgx_rdm_select(self._handle, arr, None))

    def eigenvectors_batch(self, requests):
        """requests: [(side, k, count), ...] -> list of (count, n) tensors, all gathered by ONE launch (dmrgx_rdm_eigenvectors_batch)."""
        outs, tasks = [], (_capi.RdmVecTask * max(len(requests), 1))()
        for i, (side, k, count) in enumerate(requests):
            n = self.size(side, k)
            outs.append(torch.empty((count, n), dtype=torch.float64, device="cuda"))
            tasks[i].side, tasks[i].k, tasks[i].count, tasks[i].dst_dev, tasks[i].ld = side, k, count, outs[-1].data_ptr(), n
        _capi.check(_capi.lib().dmrgx_rdm_eigenvectors_batch(self._handle, len(requests), tasks, None))
        return outs

    def eigenvectors(self, side, k, count):
        n = self.size(side, k)
        dst = torch.empty((count, n), dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _capi.check(_capi.lib().dmrgx_rdm_eigenvectors(self._handle, side, k, count, C.c_void_p(dst.data_ptr()), n, st))
        return dst

    def destroy(self):
        if self._handle:
            h, self._handle = self._handle, C.c_void_p()      # (the object is gone whatever the verdict of its verification)
            _capi.check(_capi.lib().dmrgx_rdm_destroy(h))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Communicator:
    """dmrgx_comm handle: RCCL over xGMI (one process per GPU), or the host-staged rehearsal back-end for several ranks on one
    GPU.  The 128-byte RCCL id is produced by rank 0 (Communicator.unique_id()) and handed to the other ranks by the caller
    (bench.py: one torch.distributed broadcast at start-up; the C++ engine: a rendezvous file)."""

    def __init__(self, rank, world, unique_id=None, host_staged_name=None):
        L = _capi.lib()
        self.handle = C.c_void_p()
        self.rank, self.world = rank, world
        if host_staged_name is not None:
            _capi.check(L.dmrgx_comm_init_host_staged(rank, world, host_staged_name.encode(), C.byref(self.handle)))
        else:
            assert unique_id is not None and len(unique_id) == 128
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
            _capi.check(L.dmrgx_comm_init(rank, world, C.cast(buf, C.c_void_p), C.byref(self.handle)))

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _capi.check(_capi.lib().dmrgx_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @staticmethod
    def _st(stream):
        return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)

    def allgather(self, full, seg_stride, stream=None):
        _capi.check(_capi.lib().dmrgx_comm_allgather(self.handle, C.c_void_p(full.data_ptr()), seg_stride, self._st(stream)))

    def allreduce_sum(self, buf, stream=None):
        _capi.check(_capi.lib().dmrgx_comm_allreduce_sum(self.handle, C.c_void_p(buf.data_ptr()), buf.numel(), self._st(stream)))

    def bcast(self, buf, root, stream=None):
        _capi.check(_capi.lib().dmrgx_comm_bcast(self.handle, C.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size(), root, self._st(stream)))

    def allgather_host(self, arr, stream=None):
        """arr: C-contiguous numpy array (same shape on every rank) -> array of shape (world,) + arr.shape."""
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + arr.shape, dtype=arr.dtype)
        _capi.check(_capi.lib().dmrgx_comm_allgather_host(self.handle, arr.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), arr.nbytes, self._st(stream)))
        return out

    def barrier(self, stream=None):
        _capi.check(_capi.lib().dmrgx_comm_barrier(self.handle, self._st(stream)))

    def info(self):
        """(rank, world, backend) as the library's communicator reports them (backend 0 = RCCL, 1 = host-staged)."""
        r, w, b = C.c_int32(), C.c_int32(), C.c_int32()
        _capi.check(_capi.lib().dmrgx_comm_info(self.handle, C.byref(r), C.byref(w), C.byref(b)))
        return r.value, w.value, b.value

    def destroy(self):
        if self.handle:
            _capi.check(_capi.lib().dmrgx_comm_destroy(self.handle))
            self.handle = C.c_void_p()
