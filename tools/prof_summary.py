#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/prof_net.sh / prof_bench.sh.
usage: python tools/prof_summary.py gpurun_out/prof_net [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else "k_tower"


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


vals = {}
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    f = newest(os.path.join(d, "*", "*_counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
        print("%-28s launches=%-4d mean per launch = %.6g" % (k, len(v), vals[k]))
f = newest(os.path.join(root, "trace", "*", "*_kernel_stats.csv"))
if f:
    print(open(f).read())
if "GRBM_GUI_ACTIVE" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    print("shader cycles per XCD per launch: %.4g" % cyc)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        print("MFMA pipe busy (of 1024 SIMDs x cycles): %.3f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
    if "SQ_LDS_IDX_ACTIVE" in vals:
        print("LDS busy (of 256 CUs x cycles): %.3f; bank-conflict share of LDS cycles: %.3f" % (
            vals["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), vals.get("SQ_LDS_BANK_CONFLICT", 0) / vals["SQ_LDS_IDX_ACTIVE"]))
if "SQ_INSTS_MFMA" in vals and vals["SQ_INSTS_MFMA"]:
    print("VALU instructions per MFMA: %.2f" % (vals["SQ_INSTS_VALU"] / vals["SQ_INSTS_MFMA"]))
if "SQ_WAVE_CYCLES" in vals:
    print("of wave cycles: waiting (s_waitcnt/barrier) %.3f, issue-stalled %.3f, issuing %.3f" % tuple(
        vals[k] / vals["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")))
if "FETCH_SIZE" in vals:
    # gfx950: FETCH_SIZE under-reports wide streaming reads by 2x (MI355X_MICROARCH.md §HBM); KiB units
    print("HBM traffic per launch: FETCH_SIZE %.4g KiB (x2 corrected: %.4g MB), WRITE_SIZE %.4g KiB (%.4g MB)" % (
        vals["FETCH_SIZE"], 2 * vals["FETCH_SIZE"] * 1024 / 1e6, vals.get("WRITE_SIZE", 0), vals.get("WRITE_SIZE", 0) * 1024 / 1e6))
