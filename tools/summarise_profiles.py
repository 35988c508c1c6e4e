#!/usr/bin/env python3
"""Copy the round's rocprofv3 outputs (tools/profile_round.sh -> gpurun_out/) into profiles/ as compact summaries."""
import collections, csv, json, re, shutil, subprocess, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", "gpurun_out/pmc_fetch", "gpurun_out/pmc_write", f"profiles/{R}_pmc_traffic.json", sys.argv[2] if len(sys.argv) > 2 else "16384"])
shutil.copy("gpurun_out/prof_round/p_kernel_stats.csv", f"profiles/{R}_bench_kernel_stats.csv")
shutil.copy("gpurun_out/bench_prof_b.json", f"profiles/{R}_bench_under_rocprof.json")


def short(k):
    m = re.search(r"apr::([A-Za-z0-9_]+(<[^>]*>)?)", k)
    return m.group(1) if m else k


for name, src in (("fetch_size", "gpurun_out/pmc_fetch"), ("write_size", "gpurun_out/pmc_write")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f"{src}/p_counter_collection.csv")):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])
        agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    with open(f"profiles/{R}_pmc_{name}.csv", "w") as fh:
        fh.write("Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value_sum_over_xcds\n")
        for (d, k, g, c), v in agg.items():
            fh.write(f'{d},"{short(k)}",{g},{c},{v}\n')

agg = collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/pmc_mfma/p_counter_collection.csv")):
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
    agg.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
kt = {int(r["Dispatch_Id"]): int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open("gpurun_out/pmc_mfma/p_kernel_trace.csv"))}
out = collections.OrderedDict()
for (d, k), c in agg.items():
    if c.get("GRBM_GUI_ACTIVE", 0) <= 0 or "apr::" not in k:
        continue
    a = out.setdefault(short(k), dict(launches=0, mfma_busy_cycles=0.0, gui_active=0.0, ns=0))
    a["launches"] += 1; a["mfma_busy_cycles"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); a["gui_active"] += c["GRBM_GUI_ACTIVE"]; a["ns"] += kt.get(d, 0)
for n, a in out.items():
    a["clock_GHz"] = a["gui_active"] / 8 / (a["ns"] * 1e-9) / 1e9 if a["ns"] else 0
    a["mfma_util"] = a["mfma_busy_cycles"] / a["gui_active"] / 128.0
    print(f"{n:40s} n={a['launches']:3d} clk={a['clock_GHz']:.2f} GHz  mfma util={a['mfma_util']:.3f}")
json.dump({"note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) / (GRBM_GUI_ACTIVE summed over 8 XCDs x 128 SIMDs per XCD / 8); "
                   "clock = GRBM_GUI_ACTIVE/8/duration; bench.py --files 8 --seconds 2049 (one 16384-window pass + 7-window tail)",
           "kernels": out}, open(f"profiles/{R}_pmc_mfma_busy.json", "w"), indent=1)
