#!/usr/bin/env python3
"""profiles/traffic.json entry for one workload from a tools/profile.sh output directory (gpurun_out/prof_<tag>): per-launch FETCH_SIZE x c +
WRITE_SIZE of the dominant kernel, L2 hit rate, MFMA-busy fraction, and the hash of the kernel / plan sources it was measured on (bench.py
refuses to quote it for other sources).  The factor c: MI355X_MICROARCH.md's HBM section gives x 2 for 16-byte-per-lane streaming reads and
says "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern" -- this kernel loads 8 bytes per lane
(global_load_dwordx2, 16 lanes per 128-byte row segment), so c is the ratio known / reported bytes of the known-byte run of the SAME kernel
that profile.sh makes in the same session (tools/ggemm_probe: 4.29 GB streamed from beyond the Infinity Cache; 1.49 in round 4, 1.89 for the
round-1 loader).  The x 2 figure is kept beside it for comparison with rounds 1-4 (VERDICT round 4, K1 item d).   usage: make_traffic_json.py gpurun_out/prof_r04 cfg4real [summary file name]"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
d, wl = sys.argv[1], sys.argv[2]
KERNEL = "ggemm_kernel_64"


def avg(pattern, counter, kernel=KERNEL):
    vals = []
    for f in glob.glob(os.path.join(d, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch, n = avg("pmc_FETCH_SIZE", "FETCH_SIZE")
write, _ = avg("pmc_WRITE_SIZE", "WRITE_SIZE")
hit, _ = avg("pmc_TCC_HIT*", "TCC_HIT_sum")
miss, _ = avg("pmc_TCC_HIT*", "TCC_MISS_sum")
busy, _ = avg("pmc_SQ_WAVES*", "SQ_VALU_MFMA_BUSY_CYCLES")
gui, _ = avg("pmc_SQ_WAVES*", "GRBM_GUI_ACTIVE")
cf, _ = avg("pmc_calib_FETCH_SIZE", "FETCH_SIZE", "ggemm_kernel")
cw, _ = avg("pmc_calib_WRITE_SIZE", "WRITE_SIZE", "ggemm_kernel")
stats = {}
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Name"]:
            stats = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]))
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
entry = {
    "kernel": "dmrgx::" + KERNEL,
    "bytes_per_launch": ((4294967296 / 1024.0 / cf) * fetch + write) * 1024.0,
    "bytes_per_launch_with_the_x2_of_16B_loads": (2.0 * fetch + write) * 1024.0,
    "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write,
    "correction": "FETCH_SIZE x (known / reported bytes of the same kernel's known-byte run in the same session: 8-byte-per-lane loads are an "
                  "access width MI355X_MICROARCH.md's HBM section leaves to calibration; its x2 holds for 16-byte-per-lane loads), WRITE_SIZE x1; both in KiB",
    "calibration": {"probe": "tools/ggemm_probe 1024 2048 0 0 3 (same kernel, every group its own operands)", "known_read_bytes": 4294967296,
                    "fetch_size_kib": cf, "ratio_known_over_reported": (4294967296 / 1024.0 / cf) if cf else None,
                    "known_write_bytes": 268435456, "write_size_kib": cw},
    "l2_hit_rate": hit / (hit + miss) if hit is not None and miss is not None else None,
    # MFMA pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): busy cycles summed over the chip's SIMDs over
    # the cycles the launch was active (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs)
    "mfma_busy_frac": busy / (gui / 8.0 * 1024.0) if busy and gui else None,
    "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)",
    "rocprofv3_avg_launch_ms": stats.get("avg_ns", 0) / 1e6, "rocprofv3_calls": stats.get("calls"),
    "dispatches_averaged": n,
    "source_hash": kernel_source_hash(), "measured_at_commit": commit,
    "source": "%s (rocprofv3 --kernel-trace --pmc passes of `python3 bench.py --no-sweep --no-cpu-baseline --steps 16 --warmup 4`, tools/profile.sh; per-dispatch average over stage-1 and stage-2 launches)" % (sys.argv[3] if len(sys.argv) > 3 else d),
}
tf = os.path.join(ROOT, "profiles", "traffic.json")
allt = json.load(open(tf)) if os.path.exists(tf) else {}
allt["%s@1" % wl] = entry
json.dump(allt, open(tf, "w"), indent=1)
print(json.dumps(entry, indent=1))
