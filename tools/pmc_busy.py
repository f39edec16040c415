#!/usr/bin/env python
"""MFMA-busy table of a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass (counter_collection.csv):
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per kernel whose name contains one of the given substrings.
    python tools/pmc_busy.py <counter_collection.csv> substr [substr ...]"""
import collections
import csv
import sys

path, keys = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for x in csv.DictReader(open(path)):
    for k in keys:
        if k in x["Kernel_Name"]:
            acc[k][x["Counter_Name"]] += float(x["Counter_Value"])
            disp[k].add(x["Dispatch_Id"])
            break
print("| kernel | launches | MFMA busy / SIMD-cycles |\n|---|---|---|")
for k in keys:
    if k in acc:
        gui = acc[k]["GRBM_GUI_ACTIVE"] / 8.0
        print(f"| {k} | {len(disp[k])} | {acc[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024) if gui else 0:.3f} |")
