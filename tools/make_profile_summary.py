"""Turn one tools/prof.sh run into the files kept under profiles/ (run on the GPU box, right after the passes):
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows of this repository's kernels
  <tag>_pmc_summary.txt    mean PMC counter values per kernel and dispatch (+ derived VALU instructions per wave)
  <tag>_traffic.json       HBM bytes per dispatch: FETCH_SIZE (KiB; x2 on gfx950 for wide coalesced reads, calibrated on
                           the copy probe of the same run) + WRITE_SIZE (KiB), stamped with the hash of the kernel
                           sources so that bench.py quotes it only for exactly this code
usage: python tools/make_profile_summary.py <prof dir> <tag> "<bench command>" """
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
dst = os.path.join(out_dir, "summary")
os.makedirs(dst, exist_ok=True)


def source_sha16():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "video-fingerprinting_amd", "csrc")
    for path in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, "include", "offmark_hip.h")]:
        h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]


def short(k):
    # names = bench.py's timing kinds, so that its `roofline.traffic` finds the dominant kernel of the profiled command
    for key, name in (("analyze_yuv420", "planar_analyze"), ("mark_yuv420", "planar_mark"), ("analyze_kernel", "analyze"), ("svd8_rgb8", "svd"),
                      ("mark_rgb8_kernel<true, true", "mark_fused"), ("mark_rgb8_kernel<false, true", "mark_fused"), ("mark_rgb8", "mark"),
                      ("finalize", "finalize"), ("copy16", "copy16"), ("read16", "read16"), ("svd_rgb8", "svd"), ("degenerate", "degenerate")):
        if key in k:
            return name
    return None


sha = source_sha16()
# ---- kernel stats -------------------------------------------------------------------------------------------
rows = []
for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rd = list(csv.reader(open(f)))
    rows = [rd[0]] + [r for r in rd[1:] if "ofmk::" in r[0]]
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats, kernel sources {sha}; command: {cmd}\n")
    csv.writer(f, quoting=csv.QUOTE_ALL).writerows(rows)
# ---- PMC ----------------------------------------------------------------------------------------------------------
acc = collections.defaultdict(lambda: collections.defaultdict(list))
kernel_names = {}
for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = short(r["Kernel_Name"]) if "ofmk::" in r["Kernel_Name"] else None
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernel_names.setdefault(name, r["Kernel_Name"].split("(")[0].replace("void ", ""))
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
with open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w") as f:
    f.write(f"# rocprofv3 PMC summary (MI355X, gfx950), kernel sources {sha}.  Five separate passes of:\n#   {cmd}\n"
            "# Mean counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes\n"
            "# of a wide coalesced read (MI355X_MICROARCH.md, HBM section): calibrated below on the copy probe of this run.\n")
    for name, cs in sorted(mean.items()):
        f.write(f"{name}    [{kernel_names.get(name, '')}]\n")
        for c, v in sorted(cs.items()):
            f.write(f"   {c:24s} mean {v:16.1f}  n={len(acc[name][c])}\n")
        if "SQ_INSTS_VALU" in cs and cs.get("SQ_WAVES"):
            f.write(f"   {'VALU instructions / wave':24s}      {cs['SQ_INSTS_VALU'] / cs['SQ_WAVES']:16.1f}\n")
# ---- traffic ----------------------------------------------------------------------------------------------------------
H, W, frames = 1080, 1920, 300
for tok, nxt in zip(cmd.split(), cmd.split()[1:]):
    if tok == "--frames":
        frames = int(nxt)
    if tok == "--height":
        H = int(nxt)
    if tok == "--width":
        W = int(nxt)
    if tok == "--config" and nxt == "3":
        H, W, frames = 2160, 3840, 1000
copy_bytes = frames * H * W * 3 // 16 * 16
factor = None
if "copy16" in mean and mean["copy16"].get("FETCH_SIZE"):
    factor = copy_bytes / (mean["copy16"]["FETCH_SIZE"] * 1024.0) if frames * H * W * 3 >= (1 << 28) else (1 << 30) / (mean["copy16"]["FETCH_SIZE"] * 1024.0)
traffic = {"_comment": "HBM bytes per dispatch from rocprofv3 PMC passes (tools/prof.sh): FETCH_SIZE KiB x 1024 x fetch_correction "
                       "(gfx950 half-count of wide coalesced reads, calibrated on the copy probe in the same run) + WRITE_SIZE KiB x 1024.",
           "source_sha16": sha, "command": cmd, "frames_per_dispatch": frames, "height": H, "width": W,
           "fetch_correction": round(factor, 4) if factor else 2.0, "kernel_names": kernel_names}
fc = factor if factor and 1.8 < factor < 2.2 else 2.0
for name, cs in mean.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        traffic[name] = {"fetch_bytes": int(cs["FETCH_SIZE"] * 1024 * fc), "write_bytes": int(cs["WRITE_SIZE"] * 1024)}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
print(json.dumps(traffic, indent=1))
