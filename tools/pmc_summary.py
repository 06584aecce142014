"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel name (our kernels only)."""
import csv, sys, collections, glob, os
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("analyze_kernel", "apply_", "mark_", "finalize_kernel", "copy16")): continue
        name = "analyze" if "analyze" in k else "mark_fused" if "mark_rgb8_kernel<true, true" in k else "mark" if "mark_" in k else "apply" if "apply" in k else "finalize" if "finalize" in k else "copy16" if "copy16" in k else k[:30]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
