#!/usr/bin/env python3
"""Build-time check of dmrg.x_amd/csrc/ggemm.hip: inside ggemm_kernel the staging registers v96..v127 may only be touched by the
hand-written statements (asm global loads into them, asm ds_write_b64 / v_cndmask out of them).  Between an asm load and the asm wait
the compiler believes the value has arrived, so a copy it inserted (live-range split, phi) would move stale bits -- this scan of the
generated ISA is what turns "the compiler had no reason to" into a checked property.   usage: check_staging_regs.py <file.s>"""
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
nk = 0
for m in re.finditer(r"^(_ZN5dmrgx15ggemm_kernel_64[^:\s]*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    nk += 1
    name, body = m.group(1), m.group(2)
    in_asm = False
    for line in body.split("\n"):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if not t or t.startswith(";") or t.startswith("."): continue
        regs = set()
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t): regs.update(range(int(a), int(b) + 1))
        for a in re.findall(r"\bv(\d+)\b", t): regs.add(int(a))
        if any(96 <= r <= 127 for r in regs) and not in_asm:
            print(f"{name}: compiler-generated instruction touches a staging register: {t}")
            bad += 1
if nk == 0:
    print("no ggemm_kernel found in", sys.argv[1]); sys.exit(2)
print(f"checked {nk} kernels: {'OK' if bad == 0 else str(bad) + ' violations'}")
sys.exit(1 if bad else 0)
