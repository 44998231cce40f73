"""Reads gpurun_out/trace/{stamps.npy, plan.txt}: where the workgroup-slot time of the grouped GEMM goes (tools/tile_trace.sh)."""
import sys, collections
import numpy as np
d = sys.argv[1]
st = np.load(f"{d}/stamps.npy").astype(np.int64)
plan = collections.defaultdict(list)
for l in open(f"{d}/plan.txt"):
    f = l.split()
    plan[f[0]].append(tuple(int(v) for v in f[1:]))
lists = [("s1", plan["s1"]), ("s2", plan["s2"])]
n_per_apply = sum(len(t) for _, t in lists)
reps = len(st) // n_per_apply
print(f"stamps {len(st)} = {reps} applies x {n_per_apply} workgroups")
TICK = 10.0  # ns per s_memrealtime tick (100 MHz)
off = (reps - 1) * n_per_apply          # last apply
for name, tl in lists:
    s = st[off:off + len(tl)]
    off += len(tl)
    tl = np.array(tl)
    live = tl[:, 1] >= 0
    s, tl = s[live], tl[live]
    # stamp 1 ("descriptors read, first operands issued") is only taken when a tile starts cold; a tile whose first operands were fetched
    # beside the previous tile's epilogue leaves it unset (0 or a value of an earlier launch): those two phases are then zero for it
    warm = (s[:, 1] < s[:, 0]) | (s[:, 1] > s[:, 2])
    s = s.copy()
    s[warm, 1] = s[warm, 0]
    t0 = s[:, 0].min()
    T = (s[:, :6] - t0) * TICK * 1e-3     # us
    dur = T[:, 5] - T[:, 0]
    span = T[:, 5].max()
    ks = tl[:, 6].astype(float)           # k-steps (+ scaled copies) of the tile
    M, N, tm, tn = tl[:, 4], tl[:, 5], tl[:, 2], tl[:, 3]
    mr, nr = np.minimum(64, M - tm * 64), np.minimum(64, N - tn * 64)
    fill = ((mr + 15) // 16) * ((nr + 15) // 16) / 16.0
    hw = s[:, 7] & 0xffffffff
    xcc = s[:, 7] >> 32
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"== {name}: {len(s)} tiles, launch span {span:.1f} us; slot-time sum {dur.sum()/1e3:.2f} ms = {dur.sum()/span:.0f} slots busy on average (of 1024)")
    print(f"   distinct CUs seen {len(np.unique(cuid))}, XCCs {len(np.unique(xcc))}; {int(warm.sum())} tiles started warm (first operands fetched beside the previous epilogue)")
    ph = [("descriptors", T[:, 1] - T[:, 0]), ("first operands -> LDS", T[:, 2] - T[:, 1]), ("k-step stream", T[:, 3] - T[:, 2]),
          ("scaled copies", T[:, 4] - T[:, 3]), ("epilogue stores", T[:, 5] - T[:, 4])]
    for nm, v in ph:
        print(f"   {nm:24s} mean {v.mean():7.2f} us  median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  share of slot time {v.sum()/dur.sum():.3f}")
    stream = T[:, 3] - T[:, 2]
    for cls, sel in (("full tiles", fill == 1.0), ("ragged (fill 0.5-1)", (fill < 1.0) & (fill >= 0.5)), ("slivers (fill < 0.5)", fill < 0.5)):
        if sel.sum() == 0: continue
        per = stream[sel] / np.maximum(ks[sel], 1)
        print(f"   {cls:22s} n {sel.sum():6d}  k-steps/tile {ks[sel].mean():6.1f}  stream us per k-step: mean {per.mean():.3f} median {np.median(per):.3f} p10 {np.percentile(per,10):.3f} p90 {np.percentile(per,90):.3f};  share of slot time {dur[sel].sum()/dur.sum():.3f}; overhead/tile {(dur[sel]-stream[sel]).mean():.2f} us")
    # MFMA-equivalent time: a k-step of a full tile is 16 MFMAs per wave = 1024 pipe cycles per wave, 4 waves per SIMD share the pipe
    # -> ideal us per k-step at 4 resident workgroups per CU = 4096 cycles / f
    # start-time histogram of the launch: how long until all slots are filled, and the tail
    order = np.argsort(T[:, 0])
    print(f"   start of 1024th tile {T[order[min(1023, len(order)-1)], 0]:.1f} us; tiles ending in the last 10 % of the span: {(T[:,5] > 0.9*span).sum()}; busy slots at 50 % {(((T[:,0] < 0.5*span) & (T[:,5] > 0.5*span)).sum())}, at 90 % {(((T[:,0] < 0.9*span) & (T[:,5] > 0.9*span)).sum())}, at 97 % {(((T[:,0] < 0.97*span) & (T[:,5] > 0.97*span)).sum())}")
    # per-CU occupancy over time: mean number of resident workgroups per CU
    occ = collections.Counter()
    ev = []
    # gap between consecutive workgroups on the same (CU, approx slot): estimate the dispatch gap per CU as (span*4 - sum dur on CU)
    for c in np.unique(cuid):
        sel = cuid == c
        occ[c] = dur[sel].sum() / span
    v = np.array(list(occ.values()))
    print(f"   resident workgroups per CU (time average): mean {v.mean():.2f} min {v.min():.2f} max {v.max():.2f}")
    cyc = s[:, 6].astype(float)
    okc = (cyc > 0) & (stream > 1.0)
    if okc.sum() > 10:
        ghz = cyc[okc] / (stream[okc] * 1e3)
        print(f"   shader clock during the k-step streams (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz  p10 {np.percentile(ghz,10):.3f}  p90 {np.percentile(ghz,90):.3f};  cycles per k-step of full tiles: median {np.median((cyc/np.maximum(ks,1))[okc & (fill==1.0)]):.0f}")
    # regression dur = a + b * ksteps on full tiles
    sel = fill == 1.0
    if sel.sum() > 10:
        A = np.vstack([np.ones(sel.sum()), ks[sel]]).T
        a, b = np.linalg.lstsq(A, dur[sel], rcond=None)[0]
        print(f"   full tiles: duration ~ {a:.2f} us + {b:.3f} us x k-steps")
