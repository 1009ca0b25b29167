#!/usr/bin/env python3
"""Merge the counter CSVs of separate `rocprofv3 --pmc ... --output-format csv` passes into one per-kernel summary (JSON), with
the gfx950 corrections of MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are KiB, read bytes = 2 * FETCH_SIZE * 1024
(128-B requests counted as 64 B), write bytes = WRITE_SIZE * 1024; effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration.

    python tools/pmc_summary.py OUT.json DIR_WITH_CSVS [kernel-stats.csv for durations]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n).replace("cti::(anonymous namespace)::", "").replace("cti::", "")
    n = re.sub(r"GeoF?<([^>]*)>", lambda m: "G<" + m.group(1).replace(" ", "") + ">", n)
    return re.sub(r"\(.*", "", n)


def main():
    only = None
    if "--only" in sys.argv:
        i = sys.argv.index("--only")
        only = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    out, d = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        if (os.sep + "g16pmc" in f) != (only == "gemm16"):           # the gemm16 passes ran another program: their own summary
            continue
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    dur = {}
    for f in ([] if only else glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)) + sys.argv[3:]:
        for row in csv.DictReader(open(f)):
            dur[short(row["Name"])] = float(row["AverageNs"]) / 1e3
    res = {}
    for k, cs in acc.items():
        if not k.startswith(("gemm_planes", "gemm_f16f6", "gemm16", "mbuild", "split_kernel", "quantize_f16f6", "quantize_rows", "guard_")) or (only and not k.startswith(only)):
            continue
        e = {c: v[0] / v[1] for c, v in cs.items()}
        e["calls_seen"] = max(v[1] for v in cs.values())
        if k in dur:
            e["_dur_us"] = dur[k]
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_corrected"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
        if "GRBM_GUI_ACTIVE" in e and "_dur_us" in e:
            e["effective_clock_ghz"] = round(e["GRBM_GUI_ACTIVE"] / 8 / e["_dur_us"] / 1e3, 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            # busy cycles are summed over the 1 024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: busy fraction of the launch = (busy / 1024) / (active / 8)
            e["mfma_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (e["GRBM_GUI_ACTIVE"] / 8), 4)
        res[k] = e
    json.dump({"notes": __doc__.strip().split("\n\n")[0], "kernels": res}, open(out, "w"), indent=1)
    for k, e in res.items():
        print(k, {x: e[x] for x in ("_dur_us", "hbm_read_bytes_corrected", "hbm_write_bytes", "l2_hit_rate", "effective_clock_ghz") if x in e})


if __name__ == "__main__":
    main()
