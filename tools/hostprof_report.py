"""Report for tools/hostprof.cpp samples: python3 tools/hostprof_report.py prof.txt [top]
Inclusive share per function of the repo's own modules (engine binary, libdmrgx_hip.so), and for every sample the innermost frame
as module:function (so time blocked inside the HIP runtime shows up as such)."""
import bisect, collections, os, subprocess, sys

path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45
exclude = [x for x in sys.argv[3].split(",") if x] if len(sys.argv) > 3 else []      # drop samples with one of these substrings in the stack ("the rest of the step")
include = [x for x in sys.argv[4].split(",") if x] if len(sys.argv) > 4 else []      # keep only samples with one of these substrings in the stack
maps, samples = [], []
for l in open(path):
    if l.startswith("M "):
        f = l.split()
        if len(f) < 7: continue
        lo, hi = (int(x, 16) for x in f[1].split("-"))
        maps.append((lo, hi, int(f[3], 16), f[6]))
    elif l.startswith("S"):
        samples.append([int(x, 16) for x in l.split()[1:]])
maps.sort()
base = {}
for lo, hi, off, p in maps:
    base.setdefault(p, lo - off)
syms = {}
def table(p):
    if p in syms: return syms[p]
    t = []
    try:
        out = subprocess.run(["nm", "-C", "--defined-only", "-n", p], capture_output=True, text=True).stdout
        out += subprocess.run(["nm", "-C", "-D", "--defined-only", "-n", p], capture_output=True, text=True).stdout
        for l in out.splitlines():
            f = l.split(None, 2)
            if len(f) == 3 and f[1] in "tTwW": t.append((int(f[0], 16), f[2]))
    except Exception:
        pass
    t.sort()
    syms[p] = ([a for a, _ in t], [n for _, n in t])
    return syms[p]
def resolve(a):
    i = bisect.bisect_right(maps, (a, 1 << 62, 0, "")) - 1
    if i < 0 or not (maps[i][0] <= a < maps[i][1]): return "?", "?"
    p = maps[i][3]
    rel = a - base[p]
    is_exe = p.endswith("dmrgx-square-lattice")
    addrs, names = table(p)
    for key in ((a, rel) if is_exe else (rel, a)):
        j = bisect.bisect_right(addrs, key) - 1
        if j >= 0 and key - addrs[j] < (1 << 20): return os.path.basename(p), names[j]
    return os.path.basename(p), "?"
cache = {}
def res(a):
    if a not in cache: cache[a] = resolve(a)
    return cache[a]
own = ("dmrgx-square-lattice", "libdmrgx_hip.so")
incl, leaf, deepest_own = collections.Counter(), collections.Counter(), collections.Counter()
n = ntot = 0
for s in samples:
    fr = [res(a) for a in s]
    fr = [f for f in fr if f[0] != "libhostprof.so"]
    fr = fr[1:]                                   # the signal trampoline (__restore_rt)
    if not fr: continue
    ntot += 1
    if exclude and any(x in f for _, f in fr for x in exclude): continue
    if include and not any(x in f for _, f in fr for x in include): continue
    n += 1
    leaf[fr[0][0] + ":" + fr[0][1][:70]] += 1
    seen = set()
    first = None
    for m, f in fr:
        if m in own:
            if first is None: first = f
            if f not in seen: seen.add(f); incl[f] += 1
    chain = []
    for m, f in fr:
        if m in own and (not chain or chain[-1] != f):
            chain.append(f)
            if len(chain) == 3: break
    deepest_own[" < ".join(c.split("(")[0][-48:] for c in chain) + "   [leaf in " + fr[0][0].split(".so")[0] + "]"] += 1
print(f"{n} samples" + (f" of {ntot} (excluding stacks through {exclude})" if exclude else ""))
print("== inclusive, own modules")
for f, c in incl.most_common(top): print(f"  {100*c/n:5.1f} %  {f[:150]}")
print("== innermost own function <- module of the innermost frame")
for f, c in deepest_own.most_common(top): print(f"  {100*c/n:5.1f} %  {f}")
print("== innermost frame")
for f, c in leaf.most_common(25): print(f"  {100*c/n:5.1f} %  {f}")
