"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel name (our kernels only).
usage: python tools/pmc_summary.py <dir holding pmc_* or *_sq* sub-directories>"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    tag = os.path.relpath(f, root).split(os.sep)[0].split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ofmk::" not in k:
            continue
        name = ("analyze_yuv420" if "analyze_yuv420" in k
                else "mark_yuv420" if "mark_yuv420" in k else "analyze" if "analyze_kernel" in k
                else "mark_fused" if "mark_rgb8_kernel<true, true" in k else "mark" if "mark_rgb8" in k
                else "finalize" if "finalize" in k else "copy16" if "copy16" in k else "read16" if "read16" in k
                else "svd" if "svd_rgb8" in k else k[6:36])
        acc[(tag, name)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (tag, name), cs in sorted(acc.items()):
    print(f"{tag}/{name}")
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
