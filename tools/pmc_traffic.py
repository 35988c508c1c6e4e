#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command)
into per-kernel HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- \
        python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 8 --seconds 2049
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_write ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is
taken as is (it matches the algorithmic output bytes of every kernel here exactly).  Only full
batches are kept (the run's short tail pass is dropped).
"""
import collections
import csv
import json
import re
import sys


def load(d):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]))
        agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    return agg


def short(name):
    m = re.search(r"apr::([A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    fetch, write, out = sys.argv[1:4]
    wpl = int(sys.argv[4]) if len(sys.argv) > 4 else 16384   # windows of the profiled pass (bench.py --max-batch)
    f, w = load(fetch), load(write)
    per = collections.OrderedDict()
    fl = list(f.items())
    wl = list(w.items())
    assert [k[1:] for k, _ in fl] == [k[1:] for k, _ in wl], "the two passes saw different launches"
    seen_frontend = 0
    for (k, fv), (_, wv) in zip(fl, wl):
        if "stft_mel_kernel" in k[1]:
            seen_frontend += 1
        if seen_frontend != 1:             # before the sweep, or the 1-window tail chunk of the run
            continue
        a = per.setdefault(short(k[1]), dict(launches=0, fetch_bytes=0.0, write_bytes=0.0))
        a["launches"] += 1
        a["fetch_bytes"] += 2.0 * fv * 1024.0
        a["write_bytes"] += wv * 1024.0
    for a in per.values():
        a["hbm_bytes_per_launch"] = (a["fetch_bytes"] + a["write_bytes"]) / a["launches"]
    json.dump({"windows_per_launch": wpl, "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane streams), KiB units",
               "kernels": per}, open(out, "w"), indent=1)
    for k, a in per.items():
        print(f"{k:40s} launches {a['launches']:2d}  HBM/launch {a['hbm_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
