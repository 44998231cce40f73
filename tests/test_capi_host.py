"""CPU suite: the C-ABI library loads, exports every symbol include/dmrgx.h declares, refuses to compute without a
GPU (no CPU fallback), and the host-side logic (workload generator, stripe layout incl. a world_size-2 gloo run)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "dmrgx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dmrgx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dmrgx_status"}
    assert len(declared) >= 12
    lib = pkg._capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/dmrgx.h but not exported"
    assert declared == set(pkg._capi.SIGNATURES), "ctypes binding out of sync with the header"
    assert lib.dmrgx_abi_version() == 3


def test_integration_document_prints_the_header_signatures(pkg):
    """Every call form `dmrgx_x(a, b, ...)` printed in INTEGRATION.md has the argument count of the ABI (a maintainer
    copying the document must get code that compiles)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    seen = 0
    for name, args in re.findall(r"\b(dmrgx_[a-z0-9_]+)\(([^()]*)\)", doc):
        assert name in pkg._capi.SIGNATURES, f"INTEGRATION.md names {name}, which include/dmrgx.h does not declare"
        n = 0 if not args.strip() else args.count(",") + 1
        assert n == len(pkg._capi.SIGNATURES[name][1]), f"INTEGRATION.md prints {name}({args}) but the ABI takes {len(pkg._capi.SIGNATURES[name][1])} arguments"
        seen += 1
    assert seen >= 15


def test_product_does_not_import_oracle():
    """The product package must never route through the CPU oracle."""
    for fn in os.listdir(os.path.join(ROOT, "dmrg.x_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "dmrg.x_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
    for fn in os.listdir(os.path.join(ROOT, "dmrg.x_amd", "csrc")):
        if fn.endswith((".hip", ".h", ".cpp", ".hpp")):
            assert "oracle" not in open(os.path.join(ROOT, "dmrg.x_amd", "csrc", fn)).read(), fn


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a host without a GPU")
def test_fails_loudly_without_gpu(pkg):
    with pytest.raises(pkg._capi.DmrgxError) as ei:
        pkg._capi.require_device()
    assert ei.value.code == pkg._capi.DMRGX_ERR_DEVICE and "no CPU fallback" in str(ei.value)
    from dmrgx_amd.superblock import KronPlan
    from dmrgx_amd.workloads import synthetic_superblock
    with pytest.raises(pkg._capi.DmrgxError):
        KronPlan(synthetic_superblock("cfg1", m=4, Ly=1))


def test_sector_profile_and_term_counts(pkg):
    from dmrgx_amd import workloads as wl
    assert list(wl.kept_profile(2048).values()) == [1, 9, 38, 113, 244, 388, 462, 388, 244, 113, 38, 9, 1]
    sb = wl.synthetic_superblock("cfg4")
    assert sum(sb.left_sizes) == 4096 and len(sb.terms) == 72 and len(sb.blocks) == 14
    assert sb.n_states == sum(sb.left_sizes[a] * sb.right_sizes[b] for a, b in sb.blocks)
    assert all(sb.left_qn[a] + sb.right_qn[b] == 0 for a, b in sb.blocks)
    sb5 = wl.synthetic_superblock("cfg5", m=64)
    assert len(sb5.terms) == 16            # XY: no Sz terms, no NNN (reference quirk)
    x = np.random.default_rng(0).standard_normal(wl.synthetic_superblock("cfg2", m=16, Ly=2).n_states)
    sb2 = wl.synthetic_superblock("cfg2", m=16, Ly=2)
    H = np.stack([wl.apply_factored_numpy(sb2, e) for e in np.eye(sb2.n_states)], axis=1)
    assert np.abs(H - H.T).max() < 1e-12 * np.abs(H).max()


def test_stripe_bounds_partition(pkg):
    import ctypes as C
    lib = pkg._capi.lib()
    for n in (0, 1, 5, 64, 130, 272, 850, 1028, 1693):
        for W in (1, 2, 3, 8):
            cuts = []
            for r in range(W):
                a, b = C.c_int32(), C.c_int32()
                assert lib.dmrgx_stripe_bounds(n, W, r, C.byref(a), C.byref(b)) == 0
                cuts.append((a.value, b.value))
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
            # cuts sit at the even split rounded to MFMA blocks (16 columns; the plain even split below 48 W columns), snapped to a
            # whole GEMM tile (64) where that costs no balance: every stripe is within 16 columns of its even share, the remainder of
            # n is spread instead of piling on the last rank (n = 1000, W = 8 used to give 64,128,..,168)
            g = 16 if n >= 48 * W else 1
            assert all(a % g == 0 for a, _ in cuts)
            widths = [b - a for a, b in cuts]
            assert all(abs(a - n * r // W) <= 16 for r, (a, _) in enumerate(cuts))
            assert all(abs(wd - n / W) <= 33 for wd in widths) and (n < W or min(widths) > 0), (n, W, widths)
    cuts = []
    for r in range(8):
        a, b = C.c_int32(), C.c_int32()
        assert lib.dmrgx_stripe_bounds(1000, 8, r, C.byref(a), C.byref(b)) == 0
        cuts.append(b.value - a.value)
    assert max(cuts) <= 125 + 19 and min(cuts) >= 125 - 19, cuts
    a, b = C.c_int32(), C.c_int32()
    assert lib.dmrgx_stripe_bounds(10, 2, 2, C.byref(a), C.byref(b)) == pkg._capi.DMRGX_ERR_ARG
    # per KronBlock the stripes are dealt round the ranks: rank r owns stripe (r + block) mod W, every stripe exactly once
    for W in (2, 3, 8):
        for blk in range(5):
            got = []
            for r in range(W):
                assert lib.dmrgx_stripe_bounds_of_block(1028, W, r, blk, C.byref(a), C.byref(b)) == 0
                got.append((a.value, b.value))
                c, d = C.c_int32(), C.c_int32()
                assert lib.dmrgx_stripe_bounds(1028, W, (r + blk) % W, C.byref(c), C.byref(d)) == 0 and (c.value, d.value) == got[-1]
            assert sorted(got)[0][0] == 0 and sorted(got)[-1][1] == 1028 and all(x[1] == y[0] for x, y in zip(sorted(got), sorted(got)[1:]))


def test_world_size_2_striped_apply_over_gloo():
    """N>1 path on CPU: two gloo ranks each compute their right-index stripe of y = H x (numpy, stripe rule from the
    C ABI), all-gather the rank-major segments, and rank 0 checks the result against the unstriped apply."""
    script = os.path.join(ROOT, "tests", "gloo_striped_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "striped apply ok" in outs[0]


def test_every_environment_switch_is_documented():
    """The library and the engine read a handful of environment variables (launch, memory, two solver selections that the tests
    exercise, traces).  Every one of them is named in DESIGN.md section 3's table -- an undocumented switch is an untested code path
    waiting to be forgotten (37 switches, two of them tested)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "dmrg.x_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "dmrg.x_amd", "host", "*.[ch]pp")):
        names.update(re.findall(r'getenv\("(DMRGX_[A-Z0-9_]+)"\)', open(f).read()))
    design = open(os.path.join(root, "DESIGN.md")).read()
    missing = sorted(n for n in names if n not in design)
    assert not missing, missing
    assert len(names) <= 26, sorted(names)        # (37 at the end of round 3)


def test_isa_checker_accepts_the_build_and_catches_planted_violations(tmp_path):
    """tools/check_staging_regs.py is what turns three code-generation assumptions of the dominant kernel into checked properties (`make` runs
    it on every build): here it must accept the ISA of the current build and refuse (a) a compiler-generated instruction that touches a
    staging register, (b) a tile claim written into a second register, (c) a full-tile epilogue with a store missing, (d) a spilled register."""
    import subprocess
    import sys
    isa = os.path.join(ROOT, "dmrg.x_amd", "csrc", "ggemm.isa.s")
    if not os.path.exists(isa):
        pytest.skip("no ISA listing (make has not run)")
    tool = os.path.join(ROOT, "tools", "check_staging_regs.py")
    run = lambda path: subprocess.run([sys.executable, tool, str(path)], capture_output=True, text=True)       # noqa: E731
    ok = run(isa)
    assert ok.returncode == 0, ok.stdout[-2000:]
    text = open(isa).read()
    k64 = text.index("ggemm_kernel_64")
    claim = re.search(r"global_atomic_add v(\d+), v\d+, v\d+, s\[92:93\] sc0", text[k64:])
    assert claim
    planted = {
        "staging": text.replace("s_endpgm", "v_mov_b32_e32 v100, v1\n\ts_endpgm", 1),
        "claim": text[:k64] + text[k64:].replace(claim.group(0), claim.group(0).replace("v" + claim.group(1) + ",", "v1,", 1), 1),
        "stores": text[:k64] + re.sub(r"\n\tglobal_store_dwordx2 [^\n]*\n\t;;#ASMEND", "\n\t;;#ASMEND", text[k64:], count=1),
        "spill": text.replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 3", 1),
    }
    for name, t in planted.items():
        assert t != text, name
        p = tmp_path / (name + ".s")
        p.write_text(t)
        r = run(p)
        assert r.returncode != 0, (name, r.stdout[-1500:])
