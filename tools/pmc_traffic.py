#!/usr/bin/env python3
"""HBM traffic per PGD iteration from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, "rocprofv3 PMC slots").

    tools/pmc_traffic.py <workload> <fetch_dir> <write_dir> [--into profiles/r02_traffic.json]

Units and corrections, as the guide's HBM section prescribes: both counters are in KiB (x 1024 -> bytes); on
gfx950 FETCH_SIZE reports exactly half the bytes of a coalesced streaming read -- the guide states it for 16 B per
lane; tools/probes/fetch_calib.hip (profiles/r02_fetch_calibration.txt) measured the same factor 0.5 for the 8- and
4-byte-per-lane loads the sweep kernels issue -- so the sweep kernel is doubled (its 8-B row gathers of S, served by
L2 / Infinity Cache, ride at the same factor: an upper estimate); the column-sum kernel reads short scattered runs on
16-lane pieces (uncalibrated pattern): raw in per_iteration_bytes, doubled in per_iteration_bytes_upper; WRITE_SIZE
is exact for streaming stores.  Per iteration = one launch of each of the iteration's kernels."""
import argparse
import collections
import csv
import glob
import json
import os

ITER_KERNELS = {"k_sweep_node": 2.0, "k_sweep_band": 2.0, "k_colsum_node": 1.0, "k_colsum": 1.0, "k_finalize": 1.0}


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload"); ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
    ap.add_argument("--into", default=None)
    ap.add_argument("--algorithmic", type=float, default=None)
    a = ap.parse_args()
    fetch, write = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE")
    entry, total = {}, 0.0
    # instances of the per-process warm-up problem (a handful of launches: another template instance of the same kernel) are not part of an iteration
    launches = {k: max(len(fetch.get(k, [])), len(write.get(k, []))) for k in set(fetch) | set(write)}
    most = max(launches.values(), default=0)
    for kname in sorted(set(fetch) | set(write)):
        short = kname.split("<")[0].split("(")[0].replace("void ", "").replace("desc::", "").strip()
        fac = next((v for k, v in ITER_KERNELS.items() if short.startswith(k)), None)
        if fac is None or launches[kname] * 4 < most:
            continue
        fr = sum(fetch.get(kname, [0])) / max(1, len(fetch.get(kname, [0]))) * 1024.0
        wr = sum(write.get(kname, [0])) / max(1, len(write.get(kname, [0]))) * 1024.0
        e = entry.setdefault(short, {"fetch_raw_bytes": 0.0, "write_bytes": 0.0, "traffic_bytes": 0.0, "fetch_factor": fac,
                                     "launches_sampled": len(fetch.get(kname, []))})
        e["fetch_raw_bytes"] += fr; e["write_bytes"] += wr; e["traffic_bytes"] += fac * fr + wr
        total += fac * fr + wr
    entry["per_iteration_bytes"] = total
    # upper bound: the same x2 for the column-sum kernel's short scattered runs (uncalibrated width)
    entry["per_iteration_bytes_upper"] = sum(2.0 * e["fetch_raw_bytes"] + e["write_bytes"] for e in entry.values() if isinstance(e, dict))
    if a.algorithmic:
        entry["algorithmic_bytes"] = a.algorithmic
    print(json.dumps({a.workload: entry}, indent=1))
    if a.into:
        doc = {}
        if os.path.exists(a.into):
            doc = json.load(open(a.into))
        doc.setdefault("_comment", __doc__.split("\n\n")[2].replace("\n", " "))
        doc[a.workload] = entry
        json.dump(doc, open(a.into, "w"), indent=1)


if __name__ == "__main__":
    main()
