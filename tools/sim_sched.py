"""Replay a DMRGX_PLAN_DUMP tile list through an idealised dispatcher (8 XCDs x 128 workgroup slots, in-order
dispatch to the earliest free slot) to separate scheduling loss (tails, imbalance) from kernel efficiency."""
import sys, heapq, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    f = l.split()
    rows[f[0]].append(tuple(int(x) for x in f[1:]))
c0 = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
for name, tl in rows.items():
    if not tl: continue
    slots_per_xcd = 128 if not name.endswith("b") else 64
    tot = 0.0; mk = 0.0; per = []
    work_full = 0.0; work_eff = 0.0
    for x in range(8):
        lst = [t for t in tl if t[0] % 8 == x and t[1] >= 0]
        h = [0.0] * slots_per_xcd
        heapq.heapify(h)
        for (_, g, tm, tn, M, N, cost, npr) in lst:
            t0 = heapq.heappop(h)
            heapq.heappush(h, t0 + cost + c0)
            tot += cost + c0
            mr, nr = min(64, M - tm * 64), min(64, N - tn * 64)
            work_full += cost; work_eff += cost * (((mr + 15) // 16) * ((nr + 15) // 16)) / 16.0
        per.append(max(h))
    ideal = tot / (8 * slots_per_xcd)
    ntiles = sum(1 for t in tl if t[1] >= 0)
    costs = sorted(t[6] for t in tl if t[1] >= 0)
    print(f"{name}: tiles {ntiles} (list {len(tl)}), k-steps/tile min {costs[0]} med {costs[len(costs)//2]} max {costs[-1]}; "
          f"ideal {ideal:.1f} makespan {max(per):.1f} -> sched eff {ideal/max(per):.3f}; per-XCD {[round(p) for p in per]}; mfma-block fill {work_eff/work_full:.3f}")
