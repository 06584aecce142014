"""Join the timing table of tools/bin/probe_ladder with the rocprofv3 --pmc passes of its pmc mode (tools/ladder_pmc.sh).
usage: python tools/ladder_summary.py gpurun_out/ladder_<tag>
In pmc mode the probe prints its variants in dispatch order and launches each twice, last thing in the process: the last
2 x V dispatches of every pass are the variants, in order."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
timing = {}
for line in open(os.path.join(root, "timing.txt")):
    m = re.match(r"^(.{48})\s+([\d.]+) \(([\d.]+)\)\s+([\d.]+)\s+([\d.]+) \(([\d.]+)\)\s+([\d.]+)", line)
    if m:
        timing[m.group(1).strip()] = (float(m.group(2)), float(m.group(5)))
for line in open(os.path.join(root, "timing.txt")):
    if line.startswith("#") and "identical" not in line:
        print(line.rstrip())
bad = [l for l in open(os.path.join(root, "timing.txt")) if "identical" in l and l.rstrip().endswith("NO")]
print(f"# variants whose result differs from the shipped kernels: {len(bad)}")

def load(pass_dir, log):
    names = [l[8:].strip() for l in open(log) if l.startswith("VARIANT ")]
    rows = collections.defaultdict(dict)
    kern = {}
    for f in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            kern[d] = r["Kernel_Name"]
    ids = sorted(rows)[-2 * len(names):] if names else []
    out = {}
    for i, n in enumerate(names):
        pair = ids[2 * i: 2 * i + 2]
        cs = collections.defaultdict(list)
        for d in pair:
            for c, v in rows[d].items():
                cs[c].append(v)
        out[n] = {c: sum(v) / len(v) for c, v in cs.items()}
        out[n]["_kernel"] = kern.get(pair[0], "?") if pair else "?"
    return out

for order, label in (("8", "XCD order"), ("0", "linear order")):
    merged = collections.defaultdict(dict)
    for p in ("sq", "tcc", "stall", "size"):
        d = os.path.join(root, f"o{order}_{p}")
        log = os.path.join(root, f"o{order}_{p}.log")
        if not os.path.isdir(d) or not os.path.exists(log):
            continue
        for n, cs in load(d, log).items():
            merged[n].update(cs)
    if not merged:
        continue
    print(f"\n## {label}: mean of the two launches per rung (counters summed over XCDs / SEs as rocprofv3 reports them)")
    cols = ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum",
            "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_TAG_STALL_sum", "FETCH_SIZE", "WRITE_SIZE"]
    present = [c for c in cols if any(c in v for v in merged.values())]
    print(f"{'rung':48s} {'ms':>7s} " + " ".join(f"{c.replace('_sum', '')[-16:]:>16s}" for c in present) + "   valu/wave  wait_inst/wave_cycles")
    for n, cs in merged.items():
        t = timing.get(n, (float('nan'), float('nan')))[1 if order == "8" else 0]
        vw = cs.get("SQ_INSTS_VALU", 0) / cs["SQ_WAVES"] if cs.get("SQ_WAVES") else float("nan")
        wf = cs.get("SQ_WAIT_INST_ANY", 0) / cs["SQ_WAVE_CYCLES"] if cs.get("SQ_WAVE_CYCLES") else float("nan")
        print(f"{n:48s} {t:7.4f} " + " ".join(f"{cs.get(c, float('nan')):16.0f}" for c in present) + f"   {vw:9.1f}  {wf:6.3f}")
