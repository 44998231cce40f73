#!/usr/bin/env python3
"""Build-time check of dmrg.x_amd/csrc/ggemm.hip on the generated ISA (`make`: ggemm.isa.ok).   usage: check_staging_regs.py <file.s>

1. Staging registers.  Inside ggemm_kernel_64 the registers v96..v127 may only be touched by the hand-written statements (asm global
   loads into them, asm ds_write_b64 / v_cndmask out of them).  Between an asm load and the asm wait the compiler believes the value
   has arrived, so a copy it inserted (live-range split, phi) would move stale bits -- this scan is what turns "the compiler had no
   reason to" into a checked property.
2. The tile claim.  The returning atomic of GG_CLAIM writes an ordinary C++ variable that stays in flight until the
   asm `s_waitcnt vmcnt(0)` in front of its first use.  In every kernel: the claims all write ONE register, and a compiler-generated
   instruction may only touch that register (a) to set it to the constant -1 or (b) behind an `s_waitcnt vmcnt(0)` with no claim in
   between (scanned backwards in program text).
3. The counted wait of the next tile's first operands (`vm_after`) relies on the full-tile epilogue issuing exactly TR*TC*4 = 16
   (64 x 64) / 32 (128 x 128) stores: they are asm statements, counted here.
4. Occupancy: from the code-object metadata ggemm_kernel_64 must have vgpr_count <= 128 (four waves per
   SIMD), no spilled register and no scratch; ggemm_kernel_128 likewise no spill and no scratch.
"""
import re, sys
txt = open(sys.argv[1]).read()
bad = 0


def fail(msg):
    global bad
    print(msg)
    bad += 1


def regs_of(t):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t): regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", t): regs.add(int(a))
    return regs


kernels = list(re.finditer(r"^(_ZN5dmrgx1[56]ggemm_kernel_(64|128)[^:\s]*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M))
for m in kernels:
    name, shape, body = m.group(1), m.group(2), m.group(3)
    ins = []                                      # (text, written by hand?)
    in_asm = False
    for line in body.split("\n"):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        ins.append((t.split(";")[0].strip(), in_asm))
    # 1. staging registers (the 64 x 64 kernel only: the 128 x 128 one stages through ordinary variables)
    if shape == "64":
        for t, hand in ins:
            if not hand and any(96 <= r <= 127 for r in regs_of(t)):
                fail(f"{name}: compiler-generated instruction touches a staging register: {t}")
    # 2. the claim register
    claims = [i for i, (t, hand) in enumerate(ins) if hand and t.startswith("global_atomic_add") and "s[92:93]" in t]
    if not claims:
        fail(f"{name}: no tile claim (asm global_atomic_add ... s[92:93]) found")
    creg = {re.match(r"global_atomic_add v(\d+),", ins[i][0]).group(1) for i in claims}
    if len(creg) > 1:
        fail(f"{name}: the tile claims write different registers {sorted(creg)}: the value was copied or re-defined while in flight")
    for c in creg:
        c = int(c)
        for i, (t, hand) in enumerate(ins):
            if hand or c not in regs_of(t): continue
            if re.fullmatch(rf"v_mov_b32(_e32)? v{c}, -1", t): continue
            ok = None
            for j in range(i - 1, -1, -1):
                tj, hj = ins[j]
                if j in claims: ok = False; break
                if tj.startswith("s_waitcnt") and "vmcnt(0)" in tj: ok = True; break
            if not ok:
                fail(f"{name}: compiler-generated instruction touches the claim register v{c} without an s_waitcnt vmcnt(0) since the last claim: {t}")
    # 3. stores of the full-tile epilogue: the longest run of hand-written global_store_dwordx2 with nothing but their address arithmetic between
    runs, cur = [], 0
    for t, hand in ins:
        if hand and t.startswith("global_store_dwordx2"): cur += 1
        elif hand or t.startswith(("s_cbranch", "s_branch", "s_barrier", "global_", "buffer_", "flat_", "ds_")): runs.append(cur); cur = 0
    runs.append(cur)
    want = 16 if shape == "64" else 32
    if max(runs) != want:
        fail(f"{name}: full-tile epilogue issues {max(runs)} asm stores in a row, the counted wait assumes {want}")
if len(kernels) != 2:
    fail(f"expected ggemm_kernel_64 and ggemm_kernel_128 in {sys.argv[1]}, found {len(kernels)}")

# 4. metadata
for shape, max_vgpr in (("64", 128), ("128", 128)):
    m = re.search(r"\.name:\s+_ZN5dmrgx1[56]ggemm_kernel_%s\S*\n(.*?)\.wavefront_size" % shape, txt, re.S)
    if not m:
        fail(f"no metadata for ggemm_kernel_{shape}"); continue
    md = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", m.group(1))}
    if md.get("vgpr_count", 999) > max_vgpr: fail(f"ggemm_kernel_{shape}: vgpr_count {md.get('vgpr_count')} > {max_vgpr}: fewer than four waves per SIMD")
    if md.get("vgpr_spill_count", 1) != 0: fail(f"ggemm_kernel_{shape}: {md.get('vgpr_spill_count')} spilled VGPRs")
    if md.get("private_segment_fixed_size", 1) != 0: fail(f"ggemm_kernel_{shape}: scratch in use ({md.get('private_segment_fixed_size')} bytes per lane)")
    print(f"ggemm_kernel_{shape}: {md}")
print(f"checked {len(kernels)} kernels: {'OK' if bad == 0 else str(bad) + ' violations'}")
sys.exit(1 if bad else 0)
