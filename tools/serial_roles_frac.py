#!/usr/bin/env python
"""roofline.frac recomputed from a LIW_SERIAL_ROLES=1 kernel-trace pass: the algorithmic bytes of every linearisation
of the pass (warm-up included: the stats cover the whole process) over the SUM of the role kernels' durations, instead of bench.py's
HIP-event bracket around the concurrent roles.  Writes `roofline_frac_serial_roles` into profiles/pmc_traffic.json.
usage: serial_roles_frac.py <kernel_stats.csv (tools/rocprof_summary.py)> <bench line of the same pass (json)> <pmc_traffic.json>"""
import csv
import json
import sys

stats, bench, out = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.loads(open(bench).read().strip().split("\n")[-1])
B, n, L = d["config"]["windows_per_gpu"], d["config"]["frames"], d["config"]["laser_blocks"]
passes = d["steps"] + d["warmup"]
bytes_init, bytes_marg = d["roofline"]["algorithmic_bytes_per_window"], None
bytes_marg = L * 216 + (n - 1) * (7448 + 520) + 2 * n * n * 60 + n * 120
it_mean = d["config"]["lm_iterations_mean"]
total_ms = 0.0
for r in csv.DictReader(open(stats)):
    k = r["kernel"]
    if ("k_lin_laser" in k or "k_lin_imu" in k or "k_lin_small" in k or "k_compact_active" in k) and "k_lin_all" not in k:
        total_ms += float(r["total_ms"])
alg = passes * B * ((it_mean + 1.0) * bytes_init + bytes_marg)
frac = alg / (total_ms * 1e-3) / 8e12
j = {}
try:
    j = json.load(open(out))
except Exception:
    pass
j["roofline_frac_serial_roles"] = {"frac": round(frac, 4), "role_kernel_time_ms": round(total_ms, 2), "algorithmic_bytes": alg, "passes": passes,
                                   "source": "LIW_SERIAL_ROLES=1 rocprofv3 --kernel-trace pass; sum of k_lin_laser + k_lin_imu + k_lin_small + k_compact_active"}
json.dump(j, open(out, "w"), indent=1)
print(j["roofline_frac_serial_roles"])
