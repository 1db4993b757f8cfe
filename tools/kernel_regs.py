#!/usr/bin/env python3
"""VGPR / SGPR / spill / scratch figures of every kernel of a csrc/*.hip file (cross-compiles to gfx950 assembly, reads
the .amdhsa metadata): `python tools/kernel_regs.py spmm [filter]`.  A kernel that spills into its gather loop loses more
than any scheduling idea gains -- check after every change to a hot kernel."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "spmm"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
src = os.path.join(root, "selfrec_amd", "csrc", name + ".hip")
out = f"/tmp/{name}_regs.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I" + os.path.join(root, "include"),
                "-I" + os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out] + extra, check=True,
               stderr=subprocess.DEVNULL)
text = open(out).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, flags=re.S):
    blk = m.group(0)
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]          # noqa: E731
    sym = get("name")
    dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "").replace("void ", ""))
    if flt and flt not in short:
        continue
    print(f"{short:<70} vgpr {get('vgpr_count'):>4} agpr {get('agpr_count'):>4} sgpr {get('sgpr_count'):>4} "
          f"spill v/s {get('vgpr_spill_count')}/{get('sgpr_spill_count')} scratch {get('private_segment_fixed_size'):>4} "
          f"lds {get('group_segment_fixed_size'):>6}")
