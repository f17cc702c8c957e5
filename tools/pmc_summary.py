#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

  pmc_summary.py stats <dir> <out.csv>              kernel_stats.csv with the kernel names shortened
  pmc_summary.py pmc <out.json> <name>=<dir> ...    mean counter value per (kernel, counter) over dispatches
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(attn_kernel|cent_kernel|select_position_kernel)[^ (]*", name)
    if m:
        return m.group(0)
    return name[:80]


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return hits[0] if hits else None


def stats(d, out):
    src = find(d, "*kernel_stats.csv")
    rows = list(csv.reader(open(src)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            r[0] = short(r[0])
            w.writerow(r)


def pmc(out, pairs):
    res = {}
    for pair in pairs:
        name, d = pair.split("=", 1)
        src = find(d, "*counter_collection.csv")
        acc = {}
        for r in csv.DictReader(open(src)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            v = float(r["Counter_Value"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = acc.setdefault(k, [0, 0.0, 0])
            a[0] += 1; a[1] += v; a[2] += dur
        res[name] = [{"kernel": k[0], "counter": k[1], "dispatches": a[0], "mean": a[1] / a[0],
                      "mean_duration_ns": a[2] / a[0]} for k, a in sorted(acc.items())
                     if k[0].startswith(("attn_kernel", "cent_kernel", "select_position"))]
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
