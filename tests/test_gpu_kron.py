"""Parity tests proper (-m gpu): the HIP superblock path through the C ABI vs the CPU oracle.

Tolerances: f64 throughout; the HIP path sums in a different order than the reference's row loop
(src/DMRGKron.cpp:1844-1864), so the bar is 1e-13 relative to max|y| (north_star: 1e-10 relative on E0).
"""
import os

import numpy as np
import pytest
import torch

from helpers import oracle_shell_from_superblock
from oracle.kron_c import ShellApplyC

pytestmark = pytest.mark.gpu
RTOL = 1e-13
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods(pkg):
    from dmrgx_amd import superblock, workloads, _capi
    _capi.require_device()
    return superblock, workloads, _capi


def _apply(plan, x):
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.full_like(xd, float("nan"))      # y must be overwritten, not accumulated (src/DMRGKron.cpp:1840)
    plan.apply(xd, yd)
    torch.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (64, 64, 16), (65, 63, 17), (130, 257, 100), (300, 40, 513), (17, 1, 64), (1, 200, 5)])
def test_dgemm_nn_vs_torch_fp64(mods, M, N, K):
    sbm, _, _ = mods
    g = torch.Generator(device="cpu").manual_seed(M * 1000 + N * 10 + K)
    A = torch.randn(M, K, dtype=torch.float64, generator=g).cuda()
    B = torch.randn(K, N, dtype=torch.float64, generator=g).cuda()
    C = sbm.dgemm_nn(A, B)
    ref = A @ B
    assert torch.allclose(C, ref, rtol=0, atol=1e-13 * max(1.0, float(ref.abs().max())) * K ** 0.5)


def test_dgemm_identity_asymmetric(mods):
    """A = I with an asymmetric B catches a transposed C-fragment map (guide 'Always A=I-check')."""
    sbm, _, _ = mods
    n = 80
    B = (torch.arange(n * n, dtype=torch.float64).reshape(n, n) * 1.0 + 0.25).cuda()
    C = sbm.dgemm_nn(torch.eye(n, dtype=torch.float64).cuda(), B)
    assert torch.equal(C, B)


@pytest.mark.parametrize("m,Ly,cfg", [(8, 1, "cfg1"), (16, 2, "cfg2"), (40, 3, "cfg2"), (96, 4, "cfg2"), (64, 3, "cfg3"), (70, 2, "cfg5"),
                                      # the term topologies of the headline geometries (round-2 verdict): Ly = 8 J1-J2 with its 8 NN + 16 NNN
                                      # bonds x 3 terms and the j %= Ly wrap (configs[3]), Ly = 6 Heisenberg (configs[2]), Ly = 8 XY with the
                                      # NNN bonds dropped (configs[4]); and the bench workload's own L != R sector tables, scaled down by 16
                                      (96, 8, "cfg4"), (80, 6, "cfg3"), (96, 8, "cfg5"), (0, 8, "cfg4real")])
def test_apply_matches_reference_row_loop(mods, m, Ly, cfg):
    sbm, wl, _ = mods
    if cfg == "cfg4real":
        sb = wl.synthetic_superblock(cfg, Ly=Ly, seed=77, kept=wl.scaled_real_profile("cfg4real", 16))
        assert sb.left_sizes != sb.right_sizes and len(sb.terms) == 72
    else:
        sb = wl.synthetic_superblock(cfg, m=m, Ly=Ly, seed=100 + m)
    plan = sbm.KronPlan(sb)
    ref = ShellApplyC(oracle_shell_from_superblock(sb))
    rng = np.random.default_rng(m)
    for _ in range(2):
        x = rng.standard_normal(sb.n_states)
        y, y_ref = _apply(plan, x), ref.apply(x)
        assert np.abs(y - y_ref).max() <= RTOL * np.abs(y_ref).max()
    plan.destroy()


def test_apply_medium_vs_factored_numpy(mods):
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg2", m=256, Ly=4)
    plan = sbm.KronPlan(sb)
    x = np.random.default_rng(1).standard_normal(sb.n_states)
    y, y_ref = _apply(plan, x), wl.apply_factored_numpy(sb, x)
    assert np.abs(y - y_ref).max() <= RTOL * np.abs(y_ref).max()
    plan.destroy()


def test_striped_plans_reassemble_full_apply(mods):
    """world_size 2 and 3 emulated on one GPU: each rank's stripe of y, gathered, equals the unstriped apply."""
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg2", m=60, Ly=3, seed=11)
    x = np.random.default_rng(2).standard_normal(sb.n_states)
    full = sbm.KronPlan(sb)
    y_ref = _apply(full, x)
    for W in (2, 3):
        plans = [sbm.KronPlan(sb, world_size=W, rank=r) for r in range(W)]
        info = plans[0].info
        xd = torch.from_numpy(x).cuda()
        xs = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
        plans[0].to_striped(xd, xs)
        ys = torch.zeros_like(xs)
        for r, p in enumerate(plans):
            assert p.info.seg_stride == info.seg_stride and p.info.local_offset == r * info.seg_stride
            p.apply(xs, ys[p.info.local_offset:p.info.local_offset + p.info.local_len])
        yd = torch.zeros_like(xd)
        plans[0].from_striped(ys, yd)
        torch.cuda.synchronize()
        assert np.abs(yd.cpu().numpy() - y_ref).max() <= RTOL * np.abs(y_ref).max()
        for p in plans:
            p.destroy()
    full.destroy()


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg4real", "cfg5"])
def test_full_size_properties(mods, cfg):
    """BASELINE sizes: size-independent properties -- linearity and symmetry <u,Hv> = <Hu,v>."""
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock(cfg)
    plan = sbm.KronPlan(sb)
    n = sb.n_states
    g = torch.Generator(device="cuda").manual_seed(5)
    u = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    Hu, Hv, Hc = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    plan.apply(u, Hu); plan.apply(v, Hv); plan.apply(2.0 * u - 0.5 * v, Hc)
    torch.cuda.synchronize()
    scale = float(Hu.abs().max() + Hv.abs().max())
    assert float((Hc - (2.0 * Hu - 0.5 * Hv)).abs().max()) <= 1e-12 * scale
    a, b = float(torch.dot(u, Hv)), float(torch.dot(Hu, v))
    assert abs(a - b) <= 1e-11 * float(Hu.norm() * v.norm())
    assert plan.info.n_states == n and plan.info.flops_alg > 0
    plan.destroy()


_FULL = {}


def _full_size_oracle(wl, cfg):
    """Superblock of a BASELINE config at FULL size with both CPU statements of its MatMult (built once per module: the set-up of the
    literal row loop's descriptors takes 20-90 s of host time at m = 2048 / 4096)."""
    if cfg not in _FULL:
        from oracle.kron_factored import FactoredApplyCPU
        sb = wl.synthetic_superblock(cfg)
        x = np.random.default_rng(4242).standard_normal(sb.n_states)
        y_fac = FactoredApplyCPU(sb).apply(x)
        _FULL.clear()                                   # one full-size case in host memory at a time
        _FULL[cfg] = (sb, x, y_fac, ShellApplyC(oracle_shell_from_superblock(sb)))
    return _FULL[cfg]


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5", "cfg4real"])      # (cfg4real last: the striped test below shares its oracle)
def test_full_size_apply_matches_the_oracle_numerically(mods, cfg):
    """The MatMult at the size and in the code path the bench times: thousands of tiles per launch -- resident
    workgroups claiming tiles from the per-XCD queues, the next tile's operands fetched beside the epilogue stores, split-K slabs at
    K ~ 1e4 -- against BOTH CPU statements of src/DMRGKron.cpp:1827-1869: every row against the factored form (oracle/kron_factored.py)
    and >= 4096 rows spread over all KronBlocks against the literal row loop of :1842-1864 (oracle/kron_ref.c)."""
    sbm, wl, _ = mods
    sb, x, y_fac, rows_ref = _full_size_oracle(wl, cfg)
    plan = sbm.KronPlan(sb)
    info = plan.info
    slots = 4 * 256                                     # resident workgroups of the 64 x 64 kernel (csrc/ggemm.hip: ggemm_slots)
    if cfg != "cfg3":                                   # m >= 2048: the launch must be in the claiming ("dyn") regime
        assert info.n_tiles_stage1 > slots and info.n_tiles_stage2 > slots, (info.n_tiles_stage1, info.n_tiles_stage2)
    y = _apply(plan, x)
    scale = np.abs(y_fac).max()
    # same factorisation, blocked summation on both sides: the small-size bar holds at full size
    assert np.abs(y - y_fac).max() <= RTOL * scale, np.abs(y - y_fac).max() / scale
    # the literal row loop adds the n ~ 1e7 products of a row one after the other: ITS rounding error is ~ eps sqrt(n) relative to
    # the row's magnitude sqrt(n) sigma (random-walk bound), 1.5e-13 at m = 2048 between the two CPU statements themselves -- that,
    # not 1e-13, is the bar a comparison with it can hold
    off = sb.block_offsets()
    worst, nrows = 0.0, 0
    order = sorted(range(len(sb.blocks)), key=lambda k: off[k + 1] - off[k])      # small KronBlocks first: they give all their rows
    for i, k in enumerate(order):
        n = off[k + 1] - off[k]
        take = min(n, -(-(4100 - nrows) // (len(order) - i)) + 2)
        for r0 in {off[k], off[k] + (n - take) // 2, off[k + 1] - take}:      # first, middle and last rows of every KronBlock
            c = -(-take // 3)
            r0 = int(min(r0, off[k + 1] - c))
            yr = rows_ref.apply(x, r0, r0 + c)
            prods_per_row = rows_ref.flops(r0, r0 + c) / 2.0 / c
            tol = max(RTOL, np.finfo(float).eps * prods_per_row ** 0.5)
            err = np.abs(y[r0:r0 + c] - yr[r0:r0 + c]).max() / scale
            assert err <= tol, (k, r0, err, tol)
            worst, nrows = max(worst, err), nrows + c
    assert nrows >= min(4096, sb.n_states // 2)
    # the apply must be repeatable bit for bit (fixed-order slab reduction, no atomics on data; the tile claims only decide WHO computes)
    assert np.array_equal(_apply(plan, x), y)
    plan.destroy()


@pytest.mark.parametrize("W", [2, 8])
def test_full_size_striped_apply_matches_the_oracle(mods, W):
    """Every rank's stripe of the bench's own superblock (cfg4real, full size), applied in turn on the one GPU and gathered, against the
    factored CPU statement: the plans a --gpus 2 / --gpus 8 run would build."""
    sbm, wl, _ = mods
    sb, x, y_fac, _ = _full_size_oracle(wl, "cfg4real")
    xd = torch.from_numpy(x).cuda()
    xs = ys = None
    for r in range(W):
        p = sbm.KronPlan(sb, world_size=W, rank=r)
        info = p.info
        if xs is None:
            xs = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
            p.to_striped(xd, xs)
            ys = torch.full_like(xs, float("nan"))        # every stripe must be overwritten; the padding of a segment is never read
        assert info.local_offset == r * info.seg_stride
        p.apply(xs, ys[info.local_offset:info.local_offset + info.local_len])
        if r == W - 1:
            yd = torch.zeros_like(xd)
            p.from_striped(ys, yd)
            torch.cuda.synchronize()
        p.destroy()
    y = yd.cpu().numpy()
    assert np.isfinite(y).all()
    assert np.abs(y - y_fac).max() <= RTOL * np.abs(y_fac).max()


def test_eigs_lowest_vs_dense(mods):
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg2", m=32, Ly=2, seed=3)
    plan = sbm.KronPlan(sb)
    n = sb.n_states
    H = np.stack([wl.apply_factored_numpy(sb, e) for e in np.eye(n)], axis=1)
    assert np.abs(H - H.T).max() < 1e-12 * np.abs(H).max()
    w = np.linalg.eigvalsh(H)
    e0, psi, stats = plan.eigs_lowest(tol=1e-12, seed=9)
    assert stats.converged == 1 and stats.n_matvec > 0
    assert abs(e0 - w[0]) <= 1e-10 * abs(w[0])          # north_star tolerance on E0
    r = torch.empty_like(psi)
    plan.apply(psi, r)
    assert float((r - e0 * psi).norm()) <= 1e-8 * abs(e0)
    assert abs(float(psi.norm()) - 1.0) < 1e-12
    plan.destroy()


def test_eigs_null_or_nan_start_vector_is_not_trusted(mods):
    """A caller-supplied start vector of zero norm (what the engine's projected start vectors can degenerate to) or with a NaN used to
    be normalised to the zero vector: every Krylov vector zero, the projected matrix zero, "converged" at E = 0 with psi = 0.  Both
    solver types now notice it from the first coefficients that reach the host and repeat the solve from the random start vector."""
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg2", m=32, Ly=2, seed=3)
    plan = sbm.KronPlan(sb)
    n = sb.n_states
    H = np.stack([wl.apply_factored_numpy(sb, e) for e in np.eye(n)], axis=1)
    w = np.linalg.eigvalsh(H)
    for method in (0, 1):
        for bad in ("zero", "nan"):
            psi0 = torch.zeros(plan.info.vec_len, dtype=torch.float64, device="cuda")
            if bad == "nan":
                psi0[3] = float("nan")
            e0, psi, stats = plan.eigs_lowest(tol=1e-12, seed=9, psi0=psi0, method=method)
            assert stats.converged == 1 and abs(e0 - w[0]) <= 1e-10 * abs(w[0]), (method, bad, e0, w[0])
            assert abs(float(psi.norm()) - 1.0) < 1e-12 and stats.start_rejected == 1
        # a start vector that lost most of its weight (the engine asks for |psi0|^2 >= 0.25 of its projected vectors): dropped when the
        # caller says so, used when not -- seen from the number of MatMults (the exact ground state as start converges at once)
        _, gs, ref = plan.eigs_lowest(tol=1e-12, seed=9, method=0)
        light = 0.3 * gs
        e1, _, used = plan.eigs_lowest(tol=1e-10, seed=9, psi0=light, method=method)
        e2, psi2, dropped = plan.eigs_lowest(tol=1e-10, seed=9, psi0=light, method=method, min_initial_norm2=0.25)
        e3, _, kept = plan.eigs_lowest(tol=1e-10, seed=9, psi0=0.6 * gs, method=method, min_initial_norm2=0.25)
        assert used.start_rejected == 0 and kept.start_rejected == 0 and dropped.start_rejected == 1
        assert used.n_matvec <= 4 and kept.n_matvec <= 4 and dropped.n_matvec > 10, (used.n_matvec, kept.n_matvec, dropped.n_matvec)
        for e in (e1, e2, e3):
            assert abs(e - w[0]) <= 1e-9 * abs(w[0])
        assert abs(float(psi2.norm()) - 1.0) < 1e-12
    plan.destroy()


def test_kron_diag_matches_dense_diagonal(mods):
    """dmrgx_kron_diag (the preconditioner of the generalized-Davidson option) against the diagonal of the densely assembled
    superblock Hamiltonian, unstriped and reassembled from the stripes of 3 ranks; identity cells, merged operators and both
    merge directions included."""
    import ctypes as C
    sbm, wl, capi = mods
    L = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for kw in (dict(name="cfg2", m=32, Ly=3, seed=3), dict(name="cfg1", m=12, Ly=1, seed=4), dict(name="cfg5", m=40, Ly=2, seed=5)):
        sb = wl.synthetic_superblock(kw["name"], m=kw["m"], Ly=kw["Ly"], seed=kw["seed"])
        n = sb.n_states
        H = np.stack([wl.apply_factored_numpy(sb, e) for e in np.eye(n)], axis=1)
        want = np.diag(H).copy()
        plan = sbm.KronPlan(sb)
        d = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        capi.check(L.dmrgx_kron_diag(plan._handle, C.c_void_p(d.data_ptr()), st))
        torch.cuda.synchronize()
        assert np.abs(d.cpu().numpy() - want).max() <= 1e-13 * max(1.0, np.abs(want).max())
        plan.destroy()
        W = 3
        plans = [sbm.KronPlan(sb, world_size=W, rank=r) for r in range(W)]
        info = plans[0].info
        ds = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
        for p_ in plans:
            capi.check(L.dmrgx_kron_diag(p_._handle, C.c_void_p(ds[p_.info.local_offset:].data_ptr()), st))
        dref = torch.zeros(n, dtype=torch.float64, device="cuda")
        plans[0].from_striped(ds, dref)
        torch.cuda.synchronize()
        assert np.abs(dref.cpu().numpy() - want).max() <= 1e-13 * max(1.0, np.abs(want).max())
        for p_ in plans:
            p_.destroy()


def test_eigs_generalized_davidson_vs_dense_and_lanczos(mods):
    """opts.method = 1 (SLEPc users: -H_eps_type gd): same eigenpair as the dense solve and as the Lanczos path, from a random
    start and from a perturbed eigenvector; a restart is forced by a small search space."""
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg2", m=32, Ly=2, seed=3)
    plan = sbm.KronPlan(sb)
    n = sb.n_states
    H = np.stack([wl.apply_factored_numpy(sb, e) for e in np.eye(n)], axis=1)
    w, v = np.linalg.eigh(H)
    rng = np.random.default_rng(1)
    # (without a start vector the option falls through to Lanczos; a random psi0 exercises the Davidson iteration from far away)
    # search spaces: the library's default (ncv = 0 -> 8 vectors, restarted to 1 Ritz vector + the previous one: the fused correction
    # kernel), the smallest that can grow after a restart (3), two kept Ritz vectors, and 16 (the correction kernel without the dots)
    far = lambda: torch.from_numpy(rng.standard_normal(n)).cuda()                                        # noqa: E731
    for kwargs in (dict(psi0=far(), ncv=0), dict(psi0=far(), ncv=6), dict(psi0=far(), ncv=3), dict(psi0=far(), ncv=5, gd_minv=2), dict(psi0=far()),
                   dict(psi0=torch.from_numpy(v[:, 0] + 1e-3 * rng.standard_normal(n)).cuda(), ncv=0), dict(seed=9)):
        e0, psi, stats = plan.eigs_lowest(tol=1e-12, method=1, **kwargs)
        assert stats.converged == 1 and stats.n_matvec > 0
        assert abs(e0 - w[0]) <= 1e-10 * abs(w[0])
        r = torch.empty_like(psi)
        plan.apply(psi, r)
        assert float((r - e0 * psi).norm()) <= 1e-8 * abs(e0) and abs(float(psi.norm()) - 1.0) < 1e-12
        assert abs(abs(float(psi.cpu().numpy() @ v[:, 0])) - 1.0) < 1e-8
    e_l, _, st_l = plan.eigs_lowest(tol=1e-12, seed=9)
    assert abs(e_l - e0) <= 1e-10 * abs(e0)
    plan.destroy()


def test_eigs_tiny_problem_smaller_than_ncv(mods):
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg1", m=4, Ly=1, seed=3)
    plan = sbm.KronPlan(sb)
    n = sb.n_states
    H = np.stack([wl.apply_factored_numpy(sb, e) for e in np.eye(n)], axis=1)
    e0, psi, stats = plan.eigs_lowest(tol=1e-12)
    assert abs(e0 - np.linalg.eigvalsh(H)[0]) <= 1e-10 * max(1.0, abs(e0))
    plan.destroy()


def test_bad_descriptor_is_rejected_with_reference_codes(mods):
    sbm, wl, capi = mods
    sb = wl.synthetic_superblock("cfg2", m=16, Ly=2)
    sb.blocks = sb.blocks + [sb.blocks[0]]
    with pytest.raises(capi.DmrgxError) as ei:
        sbm.KronPlan(sb)
    assert ei.value.code == capi.DMRGX_ERR_ARG
    sb = wl.synthetic_superblock("cfg2", m=16, Ly=2)
    key = next(iter(sb.left_ops))
    sb.left_ops[key].cells[0].r0 += 10_000      # planted out-of-range cell, cf. tests/UnitTests_DMRGBlock.cpp:112
    with pytest.raises(capi.DmrgxError) as ei:
        sbm.KronPlan(sb)
    assert ei.value.code == capi.DMRGX_ERR_OUTOFRANGE


@pytest.mark.parametrize("sizes", [([3, 5], [4, 2]), ([70, 1, 33], [20, 64, 65]), ([130, 40], [90, 200]), ([1100, 300], [520, 60])])
def test_rdm_spectra_and_eigenvectors_vs_lapack(mods, sizes):
    """K3/K4 vs numpy (LAPACK, as the reference's EPSLAPACK): eigenvalues of Psi Psi^T / Psi^T Psi, orthonormal
    eigenvector rows that diagonalise the block (eigenvectors are compared through invariants, their phases and
    the basis inside degenerate/null spaces are not pinned by the reference).  The last case has a rank-deficient block of
    order 1100 (rho_L of a 1100 x 60 slice): all three panel kernels of the QR preconditioner and a 1040-fold null space."""
    sbm, _, _ = mods
    ls, rs = sizes
    blocks = [(i, len(rs) - 1 - i) for i in range(min(len(ls), len(rs)))]
    rng = np.random.default_rng(sum(ls))
    psi = rng.standard_normal(sum(ls[a] * rs[b] for a, b in blocks))
    psi /= np.linalg.norm(psi)
    rdm = sbm.ReducedDensityMatrices(ls, rs, blocks, torch.from_numpy(psi).cuda())
    off = 0
    tot = [0.0, 0.0]
    for k, (a, b) in enumerate(blocks):
        Psi = psi[off:off + ls[a] * rs[b]].reshape(ls[a], rs[b])
        off += ls[a] * rs[b]
        for side, rho in ((0, Psi @ Psi.T), (1, Psi.T @ Psi)):
            w_ref = np.linalg.eigvalsh(rho)[::-1]
            w = rdm.eigenvalues(side, k)
            n = rho.shape[0]
            assert np.abs(w - w_ref).max() <= 3e-15 * n * np.abs(w_ref).max() + 1e-17      # backward-stable level: c*n*eps*||rho||
            assert np.all(np.diff(w) <= 0)
            n = rho.shape[0]
            U = rdm.eigenvectors(side, k, n).cpu().numpy()          # rows = eigenvectors
            assert np.abs(U @ U.T - np.eye(n)).max() < 1e-13
            # design tolerance of the block-Jacobi stop: off-diagonal <= 1e-12 ||rho||_F (eigenvalues are Rayleigh quotients,
            # second order in that; V stays orthogonal to round-off)
            assert np.abs(U @ rho @ U.T - np.diag(w)).max() < 1e-11 * np.linalg.norm(rho) + 1e-16
            tot[side] += w.sum()
    assert abs(tot[0] - 1.0) < 1e-13 and abs(tot[1] - 1.0) < 1e-13      # Tr rho = <psi|psi>
    rdm.destroy()


def _rdm_check(sbm, ls, rs, blocks, psi, tol_orth=1e-13):
    rdm = sbm.ReducedDensityMatrices(ls, rs, blocks, torch.from_numpy(psi).cuda())
    off = 0
    for k, (a, b) in enumerate(blocks):
        Psi = psi[off:off + ls[a] * rs[b]].reshape(ls[a], rs[b]); off += ls[a] * rs[b]
        for side, rho in ((0, Psi @ Psi.T), (1, Psi.T @ Psi)):
            n = rho.shape[0]
            w_ref = np.linalg.eigvalsh(rho)[::-1]
            w = rdm.eigenvalues(side, k)
            assert np.abs(w - w_ref).max() <= 3e-15 * n * np.abs(w_ref).max() + 1e-17, (n, side)
            U = rdm.eigenvectors(side, k, n).cpu().numpy()
            assert np.abs(U @ U.T - np.eye(n)).max() < tol_orth, (n, side)
            assert np.abs(U @ rho @ U.T - np.diag(w)).max() < 1e-11 * np.linalg.norm(rho) + 1e-16, (n, side)
    rep = {k: getattr(rdm.report, k) for k, _ in rdm.report._fields_}      # which solver path ran (dmrgx_rdm_info)
    rdm.destroy()
    return rep


def test_rdm_direct_solver_degenerate_and_boundary_cases(mods):
    """The tridiagonalisation + divide-and-conquer solver (csrc/symeig.hip) where its special paths run: exactly degenerate Schmidt
    values (type-2 deflation: Givens chains applied to the rows of the merge matrices), orders around the leaf size and the tree's
    split points (1, 2, 3, 31 .. 34, 63 .. 66), rank one, and a multiple of the identity."""
    sbm, _, _ = mods
    rng = np.random.default_rng(11)
    # degenerate pairs and a triple, graded over 12 decades
    n = 257
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.exp(-0.1 * np.arange(n)); s[1::2] = s[0::2][:len(s[1::2])]; s[10] = s[11] = s[12]
    Psi = (U * s) @ V.T
    _rdm_check(sbm, [n], [n], [(0, 0)], (Psi / np.linalg.norm(Psi)).ravel())
    # tiny orders and the boundaries of the divide-and-conquer tree, all in one call (one persistent round, many matrices)
    ls = [1, 2, 3, 31, 32, 33, 34, 63, 64, 65, 66]
    rs = list(reversed(ls))
    blocks = [(i, i) for i in range(len(ls))]
    psi = rng.standard_normal(sum(a * b for a, b in zip(ls, rs)))
    _rdm_check(sbm, ls, rs, blocks, psi / np.linalg.norm(psi))
    # rank one; and Psi = orthogonal / sqrt(n): rho = I / n (every merge deflates completely)
    u, v = rng.standard_normal(90), rng.standard_normal(70)
    P1 = np.outer(u, v)
    Q, _ = np.linalg.qr(rng.standard_normal((48, 48)))
    psi = np.concatenate([P1.ravel(), Q.ravel()])
    _rdm_check(sbm, [90, 48], [70, 48], [(0, 0), (1, 1)], psi / np.linalg.norm(psi))


def test_rdm_direct_solver_large_orders(mods):
    """Orders above the configs[3] sizes: 1500 (persistent kernel, 9 rows per workgroup) and 2200 (beyond the LDS budget of 256
    workgroups: one launch per column)."""
    sbm, _, _ = mods
    rng = np.random.default_rng(12)
    for n, r in ((1500, 700), (2200, 300)):
        psi = rng.standard_normal(n * r)
        rep = _rdm_check(sbm, [n], [r], [(0, 0)], psi / np.linalg.norm(psi), tol_orth=2e-13)
        # the path is REPORTED (dmrgx_rdm_info): persistent kernel with the rows of one matrix dealt over many workgroups, a deep merge tree and
        # dozens of WY blocks at 1500; at 2200 the larger matrix goes by launches by design, the smaller one stays persistent; no time-out
        assert rep["solver"] == 0 and rep["timed_out"] == 0 and rep["persistent_off"] == 0 and rep["merge_levels"] >= 6 and rep["wy_blocks_max"] >= 23, rep
        if n == 1500:
            assert rep["trid_persistent_matrices"] == 2 and rep["trid_launch_matrices"] == 0 and rep["max_workgroups_per_matrix"] >= 100, rep
        else:
            assert rep["trid_launch_matrices"] == 1 and rep["trid_persistent_matrices"] == 1, rep


def test_rdm_select_forms_only_the_kept_eigenvectors(mods):
    """Two-phase truncation (dmrgx_rdm_select, round 5): the spectra come first, the cut is taken on them, and only the eigenvectors of the
    kept states are formed -- the last divide-and-conquer merge, the back-transformation and the Rayleigh quotients on half-width matrices.
    Against LAPACK: the spectrum before and after the selection (the solver's own eigenvalues, unchanged), the kept eigenpairs after it
    (orthonormal rows that diagonalise rho on the kept subspace), every kept count from 0 to n incl. orders below one leaf and a
    matrix nothing is kept of; asking for more than was selected is refused."""
    sbm, _, capi = mods
    rng = np.random.default_rng(21)
    ls, rs = [300, 77, 12, 150], [120, 260, 40, 150]
    blocks = [(i, i) for i in range(4)]
    psi = rng.standard_normal(sum(a * b for a, b in zip(ls, rs)))
    psi /= np.linalg.norm(psi)
    rdm = sbm.ReducedDensityMatrices(ls, rs, blocks, torch.from_numpy(psi).cuda())
    rhos, off = [], 0
    for a, b in zip(ls, rs):
        Psi = psi[off:off + a * b].reshape(a, b); off += a * b
        rhos += [Psi @ Psi.T, Psi.T @ Psi]
    w_ref = [np.linalg.eigvalsh(r)[::-1] for r in rhos]
    for mi, r in enumerate(rhos):
        w = rdm.eigenvalues(mi % 2, mi // 2)
        assert np.abs(w - w_ref[mi]).max() <= 3e-15 * r.shape[0] * w_ref[mi][0] + 1e-17
    counts = [150, 60, 77, 0, 5, 40, 1, 150]                  # half, a part, all, nothing, below a leaf, all of a small one, one, all
    rdm.select(counts)
    for mi, r in enumerate(rhos):
        n, c = r.shape[0], counts[mi]
        w = rdm.eigenvalues(mi % 2, mi // 2)
        assert np.abs(w - w_ref[mi]).max() <= 3e-15 * n * w_ref[mi][0] + 1e-17 and np.all(np.diff(w) <= 0)
        if c:
            U = rdm.eigenvectors(mi % 2, mi // 2, c).cpu().numpy()
            assert np.abs(U @ U.T - np.eye(c)).max() < 1e-13
            assert np.abs(U @ r @ U.T - np.diag(w[:c])).max() < 1e-11 * np.linalg.norm(r) + 1e-16
            assert np.abs(r @ U.T - U.T * w[:c]).max() < 1e-11 * np.linalg.norm(r) + 1e-16          # eigenvectors of rho, not only of its restriction
        if c < n:
            with pytest.raises(capi.DmrgxError):
                rdm.eigenvectors(mi % 2, mi // 2, c + 1)
    # the same rows through the batched gather (one launch for all of them: what the engine's rotation uses)
    req = [(mi % 2, mi // 2, c) for mi, c in enumerate(counts) if c > 0]
    for (side, k, c), U in zip(req, rdm.eigenvectors_batch(req)):
        assert torch.equal(U, rdm.eigenvectors(side, k, c))
    with pytest.raises(capi.DmrgxError):
        rdm.eigenvectors_batch([(0, 0, counts[0] + 1)])
    rdm.destroy()                                              # (returns the verdict of the Rayleigh-quotient verification: raises on a mismatch)


@pytest.mark.parametrize("env", [dict(DMRGX_TRID="launch"), dict(DMRGX_RDM_SOLVER="jacobi")])
def test_rdm_alternative_paths_stay_correct(mods, env):
    """The launch-per-column tridiagonalisation (the fallback of the persistent kernel) and the round-2 block-Jacobi solver are chosen
    by process-wide switches: the LAPACK comparison in a child process with the switch set."""
    import subprocess, sys
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from __graft_entry__ import load_package; load_package();"
            "import test_gpu_kron as t; from dmrgx_amd import superblock as sbm;"
            "rng = np.random.default_rng(5); ls, rs = [300, 77], [120, 260]; psi = rng.standard_normal(300 * 120 + 77 * 260);"
            "rep = t._rdm_check(sbm, ls, rs, [(0, 0), (1, 1)], psi / np.linalg.norm(psi)); print('alt path ok', rep)") % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "alt path ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    rep = eval(p.stdout[p.stdout.index("{"):p.stdout.rindex("}") + 1])
    if "DMRGX_TRID" in env:
        assert rep["solver"] == 0 and rep["trid_launch_matrices"] == 4 and rep["trid_persistent_matrices"] == 0 and rep["timed_out"] == 0, rep
    else:
        assert rep["solver"] == 1 and rep["n_sweeps"] > 0 and rep["trid_launch_matrices"] == 0 and rep["trid_persistent_matrices"] == 0, rep


def test_persistent_tridiagonalisation_time_out_falls_back_and_stays_correct(mods):
    """The persistent LDS-resident tridiagonalisation waits for its partner workgroups with BOUNDED spins; when one never publishes (here: a
    test hook withholds one granule of the first round) every partner gives up after ~0.5 s, the status word comes back non-zero, the step is
    repeated with one launch per column and the process stops using the persistent kernel.  The spectra and eigenvectors of that very call
    -- and of a second call after the switch -- are checked against LAPACK (ADVICE / VERDICT round 3: the transition had no test)."""
    import subprocess, sys
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from __graft_entry__ import load_package; load_package();"
            "import test_gpu_kron as t; from dmrgx_amd import superblock as sbm;"
            "rng = np.random.default_rng(5); ls, rs = [300, 77], [120, 260]; psi = rng.standard_normal(300 * 120 + 77 * 260);"
            "r1 = t._rdm_check(sbm, ls, rs, [(0, 0), (1, 1)], psi / np.linalg.norm(psi)); print('first call ok');"
            "r2 = t._rdm_check(sbm, ls, rs, [(0, 0), (1, 1)], psi / np.linalg.norm(psi)); print('second call ok');"
            "assert r1['timed_out'] == 1 and r1['trid_launch_matrices'] == 4 and r1['trid_persistent_matrices'] == 0 and r1['process_timeouts'] == 1 and r1['persistent_off'] == 1, r1;"
            "assert r2['timed_out'] == 0 and r2['trid_launch_matrices'] == 4 and r2['process_timeouts'] == 1 and r2['persistent_off'] == 1, r2; print('reports ok')") % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DMRGX_TRID_FAULT="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "first call ok" in p.stdout and "second call ok" in p.stdout and "reports ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    assert "persistent tridiagonalisation timed out" in p.stderr, p.stderr[-2000:]


def _secop_from(op, keep):
    """ctypes dmrgx_secop for a workloads.SectorOperator; device tensors appended to `keep`."""
    import ctypes as C
    from dmrgx_amd import _capi
    cells = (_capi.Cell * max(len(op.cells), 1))()
    for i, c in enumerate(op.cells):
        cells[i].row_sector, cells[i].r0, cells[i].c0, cells[i].nr, cells[i].nc, cells[i].kind, cells[i].scale = c.row_sector, c.r0, c.c0, c.nr, c.nc, c.kind, c.scale
        if c.kind == 1:
            t = torch.from_numpy(np.ascontiguousarray(c.array)).cuda()
            keep.append(t)
            cells[i].data, cells[i].ld = t.data_ptr(), c.nc
    keep.append(cells)
    s = _capi.SecOp()
    s.shift, s.transposed, s.ncells, s.cells = op.shift, 0, len(op.cells), cells
    return s


def test_rdm_subset_matches_full_solve_and_refuses_unselected(mods):
    """dmrgx_rdm_create_subset (multi-GPU dealing; sides whose spectrum the engine borrows from the other side): a selected density
    matrix gets the spectrum of the full solve, an unselected one is refused with DMRGX_ERR_ARG, and the spectra of the two sides of a
    KronBlock agree up to zeros (Psi Psi^T vs Psi^T Psi) -- the identity the engine's borrowing rests on."""
    import ctypes as C
    sbm, wl, capi = mods
    L = capi.lib()
    ls, rs, blocks = [70, 33, 5], [20, 64, 41], [(0, 1), (1, 2), (2, 0)]
    n = sum(ls[a] * rs[b] for a, b in blocks)
    psi = torch.from_numpy(np.random.default_rng(5).standard_normal(n)).cuda()
    psi /= psi.norm()
    full = sbm.ReducedDensityMatrices(ls, rs, blocks, psi)
    lsz, rsz = (C.c_int32 * 3)(*ls), (C.c_int32 * 3)(*rs)
    sl, sr = capi.Sectors(3, lsz), capi.Sectors(3, rsz)
    bil, bir = (C.c_int32 * 3)(*[b[0] for b in blocks]), (C.c_int32 * 3)(*[b[1] for b in blocks])
    mask = (C.c_uint8 * 3)(1, 2, 0)                      # block 0: left only, block 1: right only, block 2: nothing
    h = C.c_void_p()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(L.dmrgx_rdm_create_subset(C.byref(sl), C.byref(sr), 3, bil, bir, C.c_void_p(psi.data_ptr()), C.cast(mask, C.c_void_p), st, C.byref(h)))
    for k, side in ((0, 0), (1, 1)):
        nn = (ls[blocks[k][0]], rs[blocks[k][1]])[side]
        out = (C.c_double * nn)()
        capi.check(L.dmrgx_rdm_eigenvalues(h, side, k, out))
        assert np.abs(np.array(out) - full.eigenvalues(side, k)).max() <= 1e-14
        other = full.eigenvalues(1 - side, k)
        m = min(len(other), nn)
        assert np.abs(np.array(out)[:m] - other[:m]).max() <= 1e-14 and np.abs(np.array(out)[m:]).max(initial=0.0) <= 1e-14
    out = (C.c_double * 70)()
    for k, side in ((0, 1), (1, 0), (2, 0), (2, 1)):
        assert L.dmrgx_rdm_eigenvalues(h, side, k, out) == capi.DMRGX_ERR_ARG
    capi.check(L.dmrgx_rdm_destroy(h))
    full.destroy()
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    capi.check(L.dmrgx_mem_stats(C.byref(a), C.byref(b), C.byref(c)))
    assert c.value > 0 and c.value >= a.value            # peak of the pool; nothing of it is in use once the handles are gone


def test_rotate_ops_vs_numpy(mods):
    """K6 vs numpy: O' = RotMatT . O . RotMat (src/DMRGBlock.cpp:766-771) for Sz-, Sp- and H-type operators with
    structurally sparse cells, an identity cell, and a sector dropped by the truncation."""
    import ctypes as C
    sbm, wl, capi = mods
    rng = np.random.default_rng(5)
    sizes = [5, 70, 33, 8, 130]
    sub = [(2, 3), (40, 30), (33, 0), (0, 8), (65, 65)]
    off = np.concatenate([[0], np.cumsum(sizes)])
    ops = [wl._old_site_op(rng, 0, sizes, sub), wl._old_site_op(rng, +1, sizes, sub), wl._sym_block_op(rng, sizes)]
    ident = wl.SectorOperator(+1, [wl.OpCell(1, 40, 0, 30, 30, wl.CELL_IDENT, 0.75), wl.OpCell(0, 0, 10, 5, 5, wl.CELL_DENSE, 0.0, rng.standard_normal((5, 5)))])
    ops.append(ident)
    old_sector, kept = [0, 1, 3, 4], [3, 41, 8, 70]           # old sector 2 is truncated away entirely
    RT = [rng.standard_normal((m, sizes[q])) for q, m in zip(old_sector, kept)]
    keep = []
    rt_dev = [torch.from_numpy(r).cuda() for r in RT]
    secops = (capi.SecOp * len(ops))(*[_secop_from(o, keep) for o in ops])
    new_of_old = {q: a for a, q in enumerate(old_sector)}
    dst, dst_arrays = [], []
    for o in ops:
        row = (C.c_void_p * len(old_sector))()
        blocks = []
        for a, q in enumerate(old_sector):
            ap = new_of_old.get(q + o.shift)
            if ap is None:
                blocks.append(None)
                continue
            t = torch.full((kept[a], kept[ap]), float("nan"), dtype=torch.float64, device="cuda")
            blocks.append(t)
            row[a] = t.data_ptr()
        dst.append(blocks)
        dst_arrays.append(row)
    dst_pp = (C.POINTER(C.c_void_p) * len(ops))(*[C.cast(r, C.POINTER(C.c_void_p)) for r in dst_arrays])
    sizes_c = (C.c_int32 * len(sizes))(*sizes)
    sec = capi.Sectors(len(sizes), sizes_c)
    rot = capi.Rotation()
    osec, kp = (C.c_int32 * 4)(*old_sector), (C.c_int32 * 4)(*kept)
    rts = (C.c_void_p * 4)(*[t.data_ptr() for t in rt_dev])
    rot.n_new, rot.old_sector, rot.kept, rot.rot_t = 4, osec, kp, rts
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(capi.lib().dmrgx_rotate_ops(C.byref(sec), C.byref(rot), len(ops), secops, dst_pp, st))
    torch.cuda.synchronize()
    for o, blocks in zip(ops, dst):
        dense = wl.operator_to_dense_blocks(o, sizes)
        for a, q in enumerate(old_sector):
            ap = new_of_old.get(q + o.shift)
            if ap is None:
                continue
            blk = dense.get(q, np.zeros((sizes[q], sizes[q + o.shift])))
            ref = RT[a] @ blk @ RT[ap].T
            got = blocks[a].cpu().numpy()
            assert np.abs(got - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max()), (o.shift, a)


def test_cells_axpy_vs_numpy(mods):
    """Dense-cell accumulate (plain and transposed, overlapping destinations applied in order) vs numpy."""
    import ctypes as C
    sbm, wl, capi = mods
    rng = np.random.default_rng(9)
    D = rng.standard_normal((90, 75))
    A, B, Cm = rng.standard_normal((40, 33)), rng.standard_normal((33, 40)), rng.standard_normal((90, 75))
    Dd, Ad, Bd, Cd = [torch.from_numpy(x.copy()).cuda() for x in (D, A, B, Cm)]
    tasks = (capi.AxpyTask * 3)()
    base = Dd.data_ptr()
    def fill(t, dst_off, src, lds, nr, nc, tr, alpha):
        t.dst, t.dst_base, t.src, t.ldd, t.lds, t.nr, t.nc, t.transposed, t.alpha = base + 8 * dst_off, base, src, 75, lds, nr, nc, tr, alpha
    fill(tasks[0], 5 * 75 + 7, Ad.data_ptr(), 33, 40, 33, 0, 2.0)           # D[5:45, 7:40] += 2 A
    fill(tasks[1], 5 * 75 + 7, Bd.data_ptr(), 40, 40, 33, 1, -0.5)          # D[5:45, 7:40] += -0.5 B^T (same destination)
    fill(tasks[2], 0, Cd.data_ptr(), 75, 90, 75, 0, 1.0)                    # D += C (overlaps both)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(capi.lib().dmrgx_cells_axpy(3, tasks, st))
    torch.cuda.synchronize()
    ref = D.copy()
    ref[5:45, 7:40] += 2.0 * A - 0.5 * B.T
    ref += Cm
    assert np.abs(Dd.cpu().numpy() - ref).max() < 1e-14


@pytest.mark.parametrize("mode", ["few_left", "few_right", "mixed_cover"])
def test_apply_when_one_side_has_fewer_distinct_operators(mods, mode):
    """Both merge directions of the plan (terms grouped by the side with fewer distinct operators, the other side's
    operators pre-summed): a mid-column cut where one site couples to every site of the other block, so the MERGED
    operator holds several cells with the same output range (O (x) 1 cell + new-site identity cell)."""
    sbm, wl, _ = mods
    sb = wl.synthetic_superblock("cfg2", m=48, Ly=3, seed=77)
    terms = []
    if mode == "mixed_cover":
        # left site 0 couples to right sites 0 and 1, left sites 1 and 2 couple to right site 2: the minimum vertex cover of the
        # term graph is {left 0, right 2} per operator type -- a left-keyed group (right operators merged) and a right-keyed
        # group (left operators merged) in one plan, 6 groups where merging on one side needs 9
        for (i, r, a) in ((0, 0, 0.7), (0, 1, 0.8), (1, 2, 0.9), (2, 2, 1.1)):
            terms += [(a, wl.OpSp, i, wl.OpSm, r), (a, wl.OpSm, i, wl.OpSp, r), (0.3 + 0.1 * i, wl.OpSz, i, wl.OpSz, r)]
    for j in range(3):
        if mode == "mixed_cover":
            break
        for i in ([2] if mode == "few_left" else [0, 1, 2]):
            jj = [j] if mode == "few_left" else [2]
            for r in jj:
                terms += [(0.7 + 0.1 * j, wl.OpSp, i, wl.OpSm, r), (0.7 + 0.1 * j, wl.OpSm, i, wl.OpSp, r), (0.3, wl.OpSz, i, wl.OpSz, r)]
    sb.terms = terms
    plan = sbm.KronPlan(sb)
    assert plan.info.n_groups == (6 if mode == "mixed_cover" else 3)
    ref = ShellApplyC(oracle_shell_from_superblock(sb))
    x = np.random.default_rng(3).standard_normal(sb.n_states)
    y, y_ref = _apply(plan, x), ref.apply(x)
    assert np.abs(y - y_ref).max() <= RTOL * np.abs(y_ref).max()
    for W in (2,):
        plans = [sbm.KronPlan(sb, world_size=W, rank=r) for r in range(W)]
        info = plans[0].info
        xs = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
        plans[0].to_striped(torch.from_numpy(x).cuda(), xs)
        ys = torch.zeros_like(xs)
        for p in plans:
            p.apply(xs, ys[p.info.local_offset:p.info.local_offset + p.info.local_len])
        yd = torch.zeros(sb.n_states, dtype=torch.float64, device="cuda")
        plans[0].from_striped(ys, yd)
        torch.cuda.synchronize()
        assert np.abs(yd.cpu().numpy() - y_ref).max() <= RTOL * np.abs(y_ref).max()
    plan.destroy()


def test_device_dot_product(pkg):
    """dmrgx_dot (VecDot of the correlator path) against numpy, fixed summation order -> bit-reproducible."""
    import ctypes as C
    import torch
    from dmrgx_amd import _capi
    L = _capi.lib()
    rng = np.random.default_rng(7)
    for n in (1, 63, 4097, 300001):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        out1, out2 = C.c_double(0.0), C.c_double(0.0)
        assert L.dmrgx_dot(n, xd.data_ptr(), yd.data_ptr(), C.byref(out1), None) == 0
        assert L.dmrgx_dot(n, xd.data_ptr(), yd.data_ptr(), C.byref(out2), None) == 0
        assert out1.value == out2.value
        assert abs(out1.value - float(x @ y)) <= 1e-12 * np.sqrt(n) * max(1.0, abs(float(x @ y)))
        dev = torch.full((3,), -1.0, dtype=torch.float64, device="cuda")      # queued form: same bits, written in stream order
        assert L.dmrgx_dot_async(n, xd.data_ptr(), yd.data_ptr(), dev.data_ptr() + 8, None) == 0
        assert dev.cpu().tolist() == [-1.0, out1.value, -1.0]
    assert L.dmrgx_dot(-1, None, None, C.byref(out1), None) == 62
    assert L.dmrgx_dot_async(5, None, None, None, None) == 62


def test_device_pool_recycles_blocks_in_stream_order(pkg):
    """dmrgx_malloc / dmrgx_free: a freed block is handed out again without touching the driver (same address for the same size
    class, a slightly smaller large request may take it too), and work queued before the free is not disturbed by the next
    owner's work queued after it (stream order)."""
    import ctypes as C
    import torch
    from dmrgx_amd import _capi
    L = _capi.lib()
    p1, p2 = C.c_void_p(), C.c_void_p()
    n = 3 << 20                                                  # 3 Mi doubles = 24 MiB
    assert L.dmrgx_malloc(C.byref(p1), n * 8) == 0 and p1.value
    host = np.arange(n, dtype=np.float64)
    assert L.dmrgx_memcpy_h2d(p1, host.ctypes.data, n * 8, None) == 0
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    assert L.dmrgx_memcpy_d2d(out.data_ptr(), p1, n * 8, None) == 0     # queued reader of the block ...
    assert L.dmrgx_free(p1) == 0                                            # ... freed while that copy may still be pending
    assert L.dmrgx_malloc(C.byref(p2), (n - 1000) * 8) == 0                # a slightly smaller request: best fit takes the cached block
    assert p2.value == p1.value
    assert L.dmrgx_memset_zero(p2, (n - 1000) * 8, None) == 0              # next owner's work, queued behind the copy
    assert np.array_equal(out.cpu().numpy(), host)
    back = np.ones(16)
    assert L.dmrgx_memcpy_d2h(back.ctypes.data, p2, 16 * 8, None) == 0 and not back.any()
    assert L.dmrgx_free(p2) == 0
    assert L.dmrgx_malloc(C.byref(p1), 0) == 0 and not p1.value            # zero bytes: null, no error


def test_dot2d_batch_matches_numpy(pkg):
    """dmrgx_dot2d_batch: strided Frobenius inner products grouped by output index (large blocks are cut into pieces),
    untouched outputs keep their value, results reproducible bit for bit."""
    import ctypes as C
    import torch
    from dmrgx_amd import _capi
    L = _capi.lib()
    rng = np.random.default_rng(11)
    shapes = [(5, 7, 0), (300, 301, 2), (64, 1, 2), (1, 900, 4), (0, 3, 5), (700, 650, 0)]
    tasks = (_capi.Dot2dTask * len(shapes))()
    keep, want = [], {}
    for i, (nr, nc, out) in enumerate(shapes):
        A, B = rng.standard_normal((nr, nc + 3)), rng.standard_normal((nr, nc + 5))
        a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        keep += [a, b]
        tasks[i].a, tasks[i].lda, tasks[i].b, tasks[i].ldb = a.data_ptr(), nc + 3, b.data_ptr(), nc + 5
        tasks[i].nr, tasks[i].nc, tasks[i].out = nr, nc, out
        want[out] = want.get(out, 0.0) + float((A[:, :nc] * B[:, :nc]).sum())
    res = []
    for _ in range(2):
        dev = torch.full((6,), -7.0, dtype=torch.float64, device="cuda")
        assert L.dmrgx_dot2d_batch(len(shapes), tasks, dev.data_ptr(), None) == 0
        res.append(dev.cpu().numpy())
    assert np.array_equal(res[0], res[1])
    for o in range(6):
        if o in want:
            assert abs(res[0][o] - want[o]) <= 1e-11 * max(1.0, abs(want[o]))
        else:
            assert res[0][o] == -7.0
    tasks[0].lda = 2
    assert L.dmrgx_dot2d_batch(len(shapes), tasks, dev.data_ptr(), None) == 62


def test_dgemm_batch_matches_numpy(pkg):
    """dmrgx_dgemm_batch: ragged independent products (and an accumulate task) in one grouped launch."""
    import ctypes as C
    import torch
    from dmrgx_amd import _capi
    L = _capi.lib()
    rng = np.random.default_rng(11)
    shapes = [(70, 33, 129), (1, 1, 1), (64, 64, 16), (200, 5, 77), (17, 140, 0)]
    host, dev, tasks = [], [], (_capi.GemmTask * len(shapes))()
    for i, (M, N, K) in enumerate(shapes):
        A, B, C0 = rng.standard_normal((M, max(K, 1))), rng.standard_normal((max(K, 1), N)), rng.standard_normal((M, N))
        acc = int(i == 3)
        Ad, Bd, Cd = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(C0.copy()).cuda()
        dev.append((Ad, Bd, Cd))
        want = (A[:, :K] @ B[:K, :]) + (C0 if acc else 0.0) if K else (C0 if acc else np.zeros((M, N)))
        host.append(want)
        tasks[i] = _capi.GemmTask(M, N, K, acc, Ad.data_ptr(), A.shape[1], Bd.data_ptr(), N, Cd.data_ptr(), N)
    assert L.dmrgx_dgemm_batch(len(shapes), C.cast(tasks, C.c_void_p), None) == 0
    for (_, _, Cd), want in zip(dev, host):
        got = Cd.cpu().numpy()
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    assert L.dmrgx_dgemm_batch(-1, None, None) == 62


def _dense_operator(op, sizes):
    off = np.concatenate([[0], np.cumsum(sizes)])
    M = np.zeros((off[-1], off[-1]))
    for c in op.cells:
        q, qc = c.row_sector, c.row_sector + op.shift
        r, s = off[q] + c.r0, off[qc] + c.c0
        if c.kind == 1:
            M[r:r + c.nr, s:s + c.nc] += c.array
        else:
            M[r:r + c.nr, s:s + c.nc] += c.scale * np.eye(c.nr, c.nc)
    return M


def test_apply_degenerate_layouts_vs_dense_kron(mods):
    """Edge cases of the layout: 1 x 1 and single-row/column KronBlocks, cells that cover only part of a sector block, an
    identity cell, terms whose shifted sector does not exist at the ends, more ranks than columns -- against the dense
    Kronecker product restricted to the target sector; a zero-size sector is rejected like the reference's
    QuantumNumbers does (src/QuantumNumbers.cpp:9-50)."""
    sbm, wl, _ = mods
    from dmrgx_amd.workloads import OpCell, SectorOperator, Superblock, CELL_DENSE, CELL_IDENT
    rng = np.random.default_rng(42)
    lsz, rsz = [1, 3, 1, 2], [2, 1, 5, 1]
    lqn = rqn = [1.5, 0.5, -0.5, -1.5]
    blocks = [(0, 3), (1, 2), (2, 1), (3, 0)]            # Sz_L + Sz_R = 0

    def dense_op(shift, sizes, partial=False):
        cells = []
        for q in range(4):
            qc = q + shift
            if not 0 <= qc < 4 or sizes[q] == 0 or sizes[qc] == 0:
                continue
            nr, nc = sizes[q], sizes[qc]
            if partial and nr > 1:                       # two cells covering the top and the bottom rows separately
                cells.append(OpCell(q, 0, 0, 1, nc, CELL_DENSE, 0.0, rng.standard_normal((1, nc))))
                cells.append(OpCell(q, 1, 0, nr - 1, nc, CELL_DENSE, 0.0, rng.standard_normal((nr - 1, nc))))
            else:
                cells.append(OpCell(q, 0, 0, nr, nc, CELL_DENSE, 0.0, rng.standard_normal((nr, nc))))
        return SectorOperator(shift, cells)

    def sym(op):
        for c in op.cells:
            if c.nr == c.nc:
                c.array = 0.5 * (c.array + c.array.T)
        return op

    OpSz_, OpSp_, OpSm_ = 0, 1, -1
    left_ops = {(OpSp_, 0): dense_op(+1, lsz, partial=True), (OpSz_, 0): dense_op(0, lsz),
                (OpSz_, 1): SectorOperator(0, [OpCell(q, 0, 0, lsz[q], lsz[q], CELL_IDENT, 0.25 * (q + 1)) for q in range(4) if lsz[q]])}
    right_ops = {(OpSp_, 0): dense_op(+1, rsz), (OpSz_, 0): dense_op(0, rsz, partial=True), (OpSz_, 1): dense_op(0, rsz)}
    terms = [(0.7, OpSp_, 0, OpSm_, 0), (0.7, OpSm_, 0, OpSp_, 0), (1.3, OpSz_, 0, OpSz_, 0), (-0.4, OpSz_, 1, OpSz_, 1)]
    sb = Superblock("edge", lsz, rsz, lqn, rqn, blocks, left_ops, right_ops, sym(dense_op(0, lsz)), sym(dense_op(0, rsz)), terms, 2, 2)
    # dense reference
    A = {k: _dense_operator(v, lsz) for k, v in left_ops.items()}
    B = {k: _dense_operator(v, rsz) for k, v in right_ops.items()}
    A[(OpSm_, 0)], B[(OpSm_, 0)] = A[(OpSp_, 0)].T, B[(OpSp_, 0)].T
    HL, HR = _dense_operator(sb.h_left, lsz), _dense_operator(sb.h_right, rsz)
    full = np.kron(HL, np.eye(sum(rsz))) + np.kron(np.eye(sum(lsz)), HR)
    for (a, io, isite, jo, jsite) in terms:
        full += a * np.kron(A[(io, isite)], B[(jo, jsite)])
    loff, roff = np.concatenate([[0], np.cumsum(lsz)]), np.concatenate([[0], np.cumsum(rsz)])
    idx = [(loff[il] + i) * sum(rsz) + roff[ir] + j for il, ir in blocks for i in range(lsz[il]) for j in range(rsz[ir])]
    Hsb = full[np.ix_(idx, idx)]
    assert sb.n_states == len(idx) == 1 + 3 * 5 + 1 + 2 * 2
    bad = Superblock("bad", [0, 3, 1, 2], rsz, lqn, rqn, blocks, {}, {}, SectorOperator(0, []), SectorOperator(0, []), [], 1, 1)
    with pytest.raises(Exception, match="size 0"):
        sbm.KronPlan(bad)
    plan = sbm.KronPlan(sb)
    for _ in range(3):
        x = rng.standard_normal(sb.n_states)
        y = _apply(plan, x)
        assert np.abs(y - Hsb @ x).max() <= 1e-13 * max(1.0, np.abs(Hsb @ x).max())
    plan.destroy()
    # more ranks than columns in some blocks: stripes of width 0 must be accepted and reassemble
    W = 4
    plans = [sbm.KronPlan(sb, world_size=W, rank=r) for r in range(W)]
    info = plans[0].info
    x = rng.standard_normal(sb.n_states)
    xs = torch.zeros(info.vec_len, dtype=torch.float64, device="cuda")
    plans[0].to_striped(torch.from_numpy(x).cuda(), xs)
    ys = torch.zeros_like(xs)
    for p in plans:
        p.apply(xs, ys[p.info.local_offset:p.info.local_offset + p.info.local_len])
    yd = torch.zeros(sb.n_states, dtype=torch.float64, device="cuda")
    plans[0].from_striped(ys, yd)
    torch.cuda.synchronize()
    assert np.abs(yd.cpu().numpy() - Hsb @ x).max() <= 1e-13 * max(1.0, np.abs(Hsb @ x).max())
    for p in plans:
        p.destroy()


def test_rdm_graded_spectrum_few_sweeps_and_warm_hints(mods):
    """Density matrices with a decaying Schmidt spectrum (like a DMRG ground state): the QR-preconditioned block Jacobi
    converges in a handful of sweeps (the unpreconditioned iteration needs 10-16 on such matrices), and
    dmrgx_rdm_create_warm accepts any orthogonal starting basis -- the eigenbasis of a nearby state or a random one --
    with the same spectra and eigenvectors, checked against LAPACK."""
    sbm, wl, _ = mods
    rng = np.random.default_rng(3)
    lsz, rsz = [40, 130, 77, 5], [64, 33, 150, 9]
    blocks = [(0, 3), (1, 2), (2, 1), (3, 0)]
    N = sum(lsz[a] * rsz[b] for a, b in blocks)
    # a state with a decaying Schmidt spectrum (like a DMRG ground state), and a small perturbation of it
    parts = []
    for a, b in blocks:
        U, _ = np.linalg.qr(rng.standard_normal((lsz[a], lsz[a])))
        V, _ = np.linalg.qr(rng.standard_normal((rsz[b], rsz[b])))
        k = min(lsz[a], rsz[b])
        s = np.exp(-0.35 * np.arange(k)) * rng.uniform(0.5, 1.0, k)
        parts.append((U[:, :k] * s) @ V[:, :k].T)
    psi0 = np.concatenate([p.ravel() for p in parts]); psi0 /= np.linalg.norm(psi0)
    psi1 = psi0 + 1e-4 * rng.standard_normal(N) * np.abs(psi0).max(); psi1 /= np.linalg.norm(psi1)
    cold0 = sbm.ReducedDensityMatrices(lsz, rsz, blocks, torch.from_numpy(psi0).cuda())
    warm = {(side, k): cold0.eigenvectors(side, k, cold0.size(side, k)).contiguous() for k in range(len(blocks)) for side in (0, 1)}
    d1 = torch.from_numpy(psi1).cuda()
    cold1 = sbm.ReducedDensityMatrices(lsz, rsz, blocks, d1)
    warm1 = sbm.ReducedDensityMatrices(lsz, rsz, blocks, d1, warm=warm)
    junk = {key: torch.from_numpy(np.linalg.qr(rng.standard_normal(tuple(t.shape)))[0]).cuda().contiguous() for key, t in warm.items()}
    junk1 = sbm.ReducedDensityMatrices(lsz, rsz, blocks, d1, warm=junk)
    # (a visit rotates the pairs between its two blocks; the pairs inside a block once per sweep -- one sweep more than with a
    # full cyclic sweep per visit, at half the latency per round)
    assert max(cold0.sweeps, cold1.sweeps, warm1.sweeps, junk1.sweeps) <= 7, (cold0.sweeps, cold1.sweeps, warm1.sweeps, junk1.sweeps)
    off = 0
    for k, (a, b) in enumerate(blocks):
        Psi = psi1[off:off + lsz[a] * rsz[b]].reshape(lsz[a], rsz[b]); off += lsz[a] * rsz[b]
        for side, rho in ((0, Psi @ Psi.T), (1, Psi.T @ Psi)):
            w_ref = np.linalg.eigvalsh(rho)[::-1]
            n = rho.shape[0]
            for r in (warm1, junk1):
                w = r.eigenvalues(side, k)
                assert np.abs(w - w_ref).max() <= 3e-15 * n * np.abs(w_ref).max() + 1e-17
                U = r.eigenvectors(side, k, n).cpu().numpy()
                assert np.abs(U @ U.T - np.eye(n)).max() < 1e-13
                assert np.abs(U @ rho @ U.T - np.diag(w)).max() < 1e-11 * np.linalg.norm(rho) + 1e-16
    for r in (cold0, cold1, warm1, junk1):
        r.destroy()
