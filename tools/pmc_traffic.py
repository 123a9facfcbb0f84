#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into
profiles/pmc_traffic.json: HBM bytes per launch of the linearise (Jacobian-evaluation) kernels.

Units / corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; WRITE_SIZE matches a known write
volume 1:1 (k_lin_laser<false>: 61 440 groups x 1 024 B = 62.91 MB measured 62.91 MB); FETCH_SIZE under-reports
coalesced streaming reads on gfx950 — calibrated here on the laser kernel, whose read volume is known exactly
(8 B per end-point component plane read — 8 planes for 2-D scans, whose four z planes the role skips, else 12 — per block + 512 B of
frame transforms per group): factor = known / reported.
usage: pmc_traffic.py <fetch_csv> <write_csv> <windows> <frames> <laser_blocks_per_window> <out_json> [planes = 8]
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
    if key == "k_marg_schur":   # since the end of round 5 two kernels on large batches (k_marg_schur_chain + k_marg_schur_eigq): their sum
        return sum(v for k, v in d.items() if key in k)
    for k, v in d.items():          # the kernel of exactly that name first (k_lin_laser_slab is a prefix of round 6's k_lin_laser_slab1)
        if key + "(" in k or k.endswith(key):
            return v
    for k, v in d.items():
        if key in k:
            return v
    return 0.0


def per_kernel_max(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: max(v) * 1024.0 for k, v in agg.items()}   # the launch with every window active


def step_known_read(B, n):
    """bytes ONE full launch of k_lm_step_quad reads (bench.py step_model; every piece is an explicit 16-byte-per-lane LDS-DMA or a
    128-bit row load, so the volume is known exactly)"""
    return B * (n * ((12 + 4 + 4) * 1024 // 4 + 3 * 120) + n * ((11 * 1024 + 6 * 256) // 4) + 8 * 128 + 2 * n * 120)   # (bench.py step_model)


def main():
    fetch_csv, write_csv, B, n, L, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    planes = int(sys.argv[7]) if len(sys.argv) > 7 else 8   # bench.py's synthetic scans are 2-D (z = 0): the laser role reads 8 of the 12 planes
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    fx, wx = per_kernel_max(fetch_csv, "FETCH_SIZE"), per_kernel_max(write_csv, "WRITE_SIZE")
    # ---- per-kernel calibration of FETCH_SIZE.  The guide's prior for 16-byte-per-lane streaming reads is x2; the
    # factor depends on the access pattern, so it is fixed per kernel on a read volume that is known exactly:
    #   k_lin_laser<true>: 8 B x planes of end points per block + 512 B of frame states per group (8-byte-per-lane SoA loads)
    #   k_lm_step_quad:    step_known_read() (LDS-DMA dwordx4 pieces + 128-bit record rows), on its launch with every window active
    #   k_lin_laser_slab (round 4, large 2-D batches): the packed rows (64 B per block for groups of equal length, as in bench.py's windows) +
    #                      two poses, the group range and the has_match byte per (window, frame)
    laser_name = "k_lin_laser_slab" if pick(fx, "k_lin_laser_slab") else "k_lin_laser<true>"
    known_laser_read = B * L * 8.0 * planes + B * n * (112.0 if laser_name == "k_lin_laser_slab" else 512.0)
    cal_laser = known_laser_read / pick(fx, laser_name) if pick(fx, laser_name) else None
    step_name = "k_lm_step_quad" if pick(fx, "k_lm_step_quad") else "k_lm_step"
    cal_step = step_known_read(B, n) / pick(fx, step_name) if step_name == "k_lm_step_quad" else cal_laser
    names = ["k_frame_tf", laser_name, "k_lin_imu", "k_lin_small"]
    kern = {}
    for k in names + [step_name, "k_marg_schur"]:
        cal = cal_step if k == step_name else cal_laser
        kern[k] = {"fetch_reported": pick(f, k), "fetch_calibrated": pick(f, k) * cal, "write": pick(w, k), "calibration": cal,
                   "fetch_reported_full_launch": pick(fx, k), "write_full_launch": pick(wx, k)}
    # ---- consistency (same item): a consumer cannot read more than its producers wrote plus what it reads of its own.  The step kernel
    # reads the partial sums the three role kernels wrote, its own record (written in the same launch) and ~25 kB of states / scales per window
    produced = sum(pick(wx, k) for k in names) + pick(wx, step_name) + B * 25e3
    step_read_full = pick(fx, step_name) * cal_step
    ok = step_read_full <= 1.10 * produced
    res = {"windows": B, "frames": n, "laser_blocks": L, "laser_planes_read": planes, "fetch_calibration_factor": cal_laser, "fetch_calibration_factor_step": cal_step,
           "step_kernel": step_name, "laser_kernel": laser_name,
           "k_linearize_hbm_bytes_per_launch": sum(kern[k]["fetch_calibrated"] + kern[k]["write"] for k in names),
           "k_lm_step_hbm_bytes_per_launch": kern[step_name]["fetch_calibrated"] + kern[step_name]["write"], "kernels": kern,
           "step_read_vs_produced": {"step_read_full_launch": step_read_full, "producers_wrote_plus_own": produced, "consistent": ok},
           "note": "FETCH_SIZE/WRITE_SIZE in KiB from separate --pmc passes; *_per_launch = means over the launches of one bench step (launches late "
                   "in a solve carry fewer active windows), *_full_launch = the launch with every window active; FETCH_SIZE calibrated PER KERNEL"}
    res["k_linearize_hbm_bytes_per_window"] = res["k_linearize_hbm_bytes_per_launch"] / B
    if not ok:
        raise SystemExit("pmc_traffic: the step kernel's calibrated read volume %.3e exceeds what its producers wrote + its own %.3e — "
                         "calibration does not transfer, refusing to write %s" % (step_read_full, produced, out))
    if cal_step and not (1.0 <= cal_step <= 3.0):
        raise SystemExit("pmc_traffic: implausible FETCH_SIZE factor %.2f for %s (guide: ~x2 for 16-byte-per-lane reads)" % (cal_step, step_name))
    prev = {}
    try:
        prev = json.load(open(out))
    except Exception:
        pass
    for k in ("k_lm_step_issue_stats", "role_issue_stats", "roofline_frac_serial_roles"):   # filled by other passes: keep
        if k in prev and k not in res:
            res[k] = prev[k]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("fetch_calibration_factor", "fetch_calibration_factor_step", "k_linearize_hbm_bytes_per_window", "k_linearize_hbm_bytes_per_launch", "k_lm_step_hbm_bytes_per_launch", "step_read_vs_produced")}))


if __name__ == "__main__":
    main()
