#!/usr/bin/env python
"""Issue statistics of the role kernels and the step kernel from the two SQ counter passes of tools/pmc_stall_passes.sh
(gpurun_out/pmc_st.csv: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY;
gpurun_out/pmc_st2.csv: ... SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES) -> `role_issue_stats` / `k_lm_step_issue_stats` of a pmc_traffic.json.
Shares are of a wave's resident cycles (SQ_WAVE_CYCLES), summed over the launches of one bench step.
usage: pmc_issue_stats.py <pmc_st.csv> <pmc_st2.csv> <pmc_traffic.json> <tag>"""
import collections
import csv
import json
import sys


def sums(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]][r["Counter_Name"]] += 1
    return agg, n


def pick(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return {}


def main():
    st, st2, out, tag = sys.argv[1:5]
    a, na = sums(st)
    b, nb = sums(st2)
    res = {}
    laser_key = "k_lin_laser_slab" if pick(a, "k_lin_laser_slab") else "k_lin_laser<true>"   # (large 2-D batches run the lane-per-group kernel)
    for key, name in ((laser_key, "k_lin_laser"), ("k_lin_imu", "k_lin_imu"), ("k_lin_small", "k_lin_small"), ("k_lm_step_quad", "k_lm_step_quad")):
        s, s2 = pick(a, key), pick(b, key)
        launches = max(pick(na, key).values()) if pick(na, key) else 0
        wc = s.get("SQ_WAVE_CYCLES", 0.0)
        if not wc:
            continue
        res[name] = {"wait_any_frac": round(s.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wait_inst_frac": round(s.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                     "active_inst_frac": round(s.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3), "valu_active_frac": round(s.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3),
                     "busy_cycles_per_launch": s.get("SQ_BUSY_CYCLES", 0.0) / max(launches, 1),
                     "mfma_insts_per_launch": s2.get("SQ_INSTS_MFMA", 0.0) / max(launches, 1),
                     "mfma_busy_cycles_per_launch": s2.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(launches, 1), "launches": launches}
    j = json.load(open(out))
    note = "share of a wave's resident cycles (SQ_WAVE_CYCLES): parked on s_waitcnt / issue stall / issuing / issuing VALU; rocprofv3 --pmc passes of tools/pmc_stall_passes.sh, %s" % tag
    step = res.pop("k_lm_step_quad", None)
    if step:
        step["note"] = note
        j["k_lm_step_issue_stats"] = step
    res["note"] = note
    j["role_issue_stats"] = res
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps({"role_issue_stats": res, "k_lm_step_issue_stats": step}, indent=1))


if __name__ == "__main__":
    main()
