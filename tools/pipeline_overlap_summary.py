"""Overlap of uploads, kernels and downloads in the plugin pipeline, from a rocprofv3 trace:
  cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -o t -- python3 tools/plugin_pipeline_rate.py 600 50 rgb24
  python3 tools/pipeline_overlap_summary.py OUT > profiles/r3_plugin_pipeline_overlap.txt
Busy time per category (union of its intervals), pairwise and three-way concurrency, over one Embedder run (the last, warm one: its cluster of
downloads; the Extractor only uploads)."""
import csv
import glob
import os
import sys


def load(pattern, start_key, end_key, name_key):
    rows = []
    for path in glob.glob(pattern, recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r[start_key]), int(r[end_key]), r.get(name_key, "")))
    return rows


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(u):
    return sum(b - a for a, b in u)


def intersect(u, v):
    i = j = 0
    out = []
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b:
            out.append([a, b])
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return out


def clip(u, lo, hi):
    return [[max(a, lo), min(b, hi)] for a, b in u if min(b, hi) > max(a, lo)]


root = sys.argv[1]
kern = load(os.path.join(root, "**", "*kernel_trace.csv"), "Start_Timestamp", "End_Timestamp", "Kernel_Name")
cop = load(os.path.join(root, "**", "*memory_copy_trace.csv"), "Start_Timestamp", "End_Timestamp", "Direction")
big = [c for c in cop if c[1] - c[0] > 200_000]                       # frame batches, not the small control copies
h2d = union([(a, b) for a, b, d in big if "HOST_TO_DEVICE" in d.upper() or d.upper().startswith("H2D")])
d2h = union([(a, b) for a, b, d in big if "DEVICE_TO_HOST" in d.upper() or d.upper().startswith("D2H")])
if not d2h:        # on this stack a device -> page-locked host copy is a blit KERNEL (__amd_rocclr_copyBuffer), not an SDMA record
    d2h = union([(a, b) for a, b, n in kern if "rocclr_copyBuffer" in n and b - a > 200_000])
k = union([(a, b) for a, b, n in kern if "ofmk" in n])
if not h2d or not d2h:
    print("no large copies in both directions found; directions seen:", sorted({d for _, _, d in cop}))
    sys.exit(0)
# one Embedder run = one cluster of downloads (the Extractor only uploads; runs are tens of ms apart): take the LAST cluster
# (warm) and start the window at the first upload that belongs to it
clusters = [[d2h[0]]]
for iv in d2h[1:]:
    if iv[0] - clusters[-1][-1][1] > 30_000_000:
        clusters.append([iv])
    else:
        clusters[-1].append(iv)
last = clusters[-1]
prev_end = clusters[-2][-1][1] if len(clusters) > 1 else 0
ups = [iv for iv in h2d if iv[1] > prev_end and iv[0] < last[-1][1]]
ups = [iv for iv in ups if last[0][0] - iv[0] < 60_000_000]             # not the Extractor run that came before it
lo, hi = min(ups[0][0], last[0][0]), last[-1][1]
h, d, kk = clip(h2d, lo, hi), clip(d2h, lo, hi), clip(k, lo, hi)
span = hi - lo
ms = lambda x: f"{x / 1e6:9.2f} ms ({100 * x / span:5.1f} % of the window)"      # noqa: E731
print(f"one Embedder.start() over the traced frames, first upload to last download: {span / 1e6:.2f} ms ({len(last)} download batches)")
print("  uploads busy          ", ms(length(h)))
print("  downloads busy        ", ms(length(d)))
print("  kernels busy          ", ms(length(kk)))
print("  uploads & downloads   ", ms(length(intersect(h, d))))
print("  uploads & kernels     ", ms(length(intersect(h, kk))))
print("  downloads & kernels   ", ms(length(intersect(d, kk))))
print("  all three at once     ", ms(length(intersect(intersect(h, d), kk))))
print("  sum of the three busy times / window = %.2f (1.0 = no overlap at all)" % ((length(h) + length(d) + length(kk)) / span))
