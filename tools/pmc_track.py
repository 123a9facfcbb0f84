#!/usr/bin/env python
"""HBM traffic of the kernels of the batched TRACK leg from two rocprofv3 --pmc passes of tools/track_batch_probe.py (FETCH_SIZE, WRITE_SIZE;
separate runs, --kernel-trace only) -> profiles/pmc_track.json.  Units / corrections as MI355X_MICROARCH.md (HBM) prescribes: the counters are
in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads -> x2 (the guide's flat factor; the C2 passes of
tools/pmc_traffic.py calibrate it per kernel at 1.9 - 2.0); WRITE_SIZE 1:1.  Per kernel: mean and MAX over the launches (max = a launch with
every window active; the LM tail launches touch almost nothing).
usage: pmc_track.py <fetch_csv> <write_csv> <robots> <out_json>"""
import collections, csv, json, sys


def agg(path, counter):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
    return a


def main():
    f, w, B, out = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3]), sys.argv[4]
    res = {"robots": B, "fetch_correction": 2.0, "note": __doc__.split("usage")[0].strip(), "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if "liw" not in k:
            continue
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        res["kernels"][k] = {"launches": len(fv), "fetch_reported_mean": sum(fv) / len(fv), "fetch_reported_max": max(fv), "write_mean": sum(wv) / len(wv), "write_max": max(wv),
                             "hbm_bytes_full_launch": 2.0 * max(fv) + max(wv), "hbm_bytes_per_robot_full_launch": (2.0 * max(fv) + max(wv)) / B}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-60s launches %4d  full launch: fetch x2 %8.1f MB + write %8.1f MB = %7.1f B per robot" % (k[:60], v["launches"], 2 * v["fetch_reported_max"] / 1e6, v["write_max"] / 1e6, v["hbm_bytes_per_robot_full_launch"]))


if __name__ == "__main__":
    main()
