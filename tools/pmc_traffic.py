#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into
profiles/pmc_traffic.json: HBM bytes per launch of the linearise (Jacobian-evaluation) kernels.

Units / corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; WRITE_SIZE matches a known write
volume 1:1 (k_lin_laser<false>: 61 440 groups x 1 024 B = 62.91 MB measured 62.91 MB); FETCH_SIZE under-reports
coalesced streaming reads on gfx950 — calibrated here on the laser kernel, whose read volume is known exactly
(96 B of end-points per block + 512 B of frame transforms per group): factor = known / reported.
usage: pmc_traffic.py <fetch_csv> <write_csv> <windows> <frames> <laser_blocks_per_window> <out_json>
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}   # KiB -> bytes, mean per launch


def pick(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return 0.0


def main():
    fetch_csv, write_csv, B, n, L, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    known_laser_read = B * L * 96.0 + B * n * 512.0
    cal = known_laser_read / pick(f, "k_lin_laser<true>")
    names = ["k_frame_tf", "k_lin_laser<true>", "k_lin_imu", "k_lin_small"]
    kern = {k: {"fetch_reported": pick(f, k), "fetch_calibrated": pick(f, k) * cal, "write": pick(w, k)} for k in names + ["k_lm_step", "k_marg_schur"]}
    lin = sum(kern[k]["fetch_calibrated"] + kern[k]["write"] for k in names)
    res = {"windows": B, "frames": n, "laser_blocks": L, "fetch_calibration_factor": cal,
           "k_linearize_hbm_bytes_per_launch": lin, "k_linearize_hbm_bytes_per_window": lin / B,
           "k_lm_step_hbm_bytes_per_launch": kern["k_lm_step"]["fetch_calibrated"] + kern["k_lm_step"]["write"], "kernels": kern,
           "note": "FETCH_SIZE/WRITE_SIZE in KiB from separate --pmc passes; means over the launches of one bench step (launches late in "
                   "a solve carry fewer active windows)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("fetch_calibration_factor", "k_linearize_hbm_bytes_per_window", "k_linearize_hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
