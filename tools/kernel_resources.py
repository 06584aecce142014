"""Registers / scratch / LDS of every kernel in the library, from a device-only assembly listing.
usage: python tools/kernel_resources.py [extra hipcc flags...]   (writes /tmp/offmark_kernels.s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

flags = [f for f in ge.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
out = "/tmp/offmark_kernels.s"
subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", os.path.join(ge.CSRC, "offmark_kernels.hip"), "-o", out,
                *sys.argv[1:]], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
for b in re.split(r"\n  - \.agpr_count:", text)[1:]:
    get = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)  # noqa: E731
    name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("ofmk::", "").replace("void ", "")
    print(f"{name:60s} vgpr {get('vgpr_count'):>3}  sgpr {get('sgpr_count'):>3}  scratch {get('private_segment_fixed_size'):>3}  lds {get('group_segment_fixed_size')}")
