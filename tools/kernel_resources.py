#!/usr/bin/env python3
"""Registers, spills, LDS and scratch of every kernel in libpptoas_hip.so, read from
the code object's metadata (no GPU needed):  python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = os.path.join(ROOT, "pulseportraiture_amd", "csrc", "libpptoas_hip.so")
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True,
                               check=True).stdout
    rows = []
    for e in re.split(r"\n  - ", notes):
        m = re.search(r"\.name:\s+(\S+)", e)
        if not m or ".vgpr_count" not in e:
            continue

        def g(k):
            mm = re.search(r"\.%s:\s+(\d+)" % k, e)
            return int(mm.group(1)) if mm else -1
        rows.append((m.group(1), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                     g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True,
                           text=True).stdout.split("\n")
    print("%-84s %5s %5s %5s %6s %6s %7s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "lds", "scratch"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void pp::", "", n)
        n = re.sub(r"\(.*\)$", "", n)
        if pat and pat not in n:
            continue
        print("%-84s %5d %5d %5d %6d %6d %7d %7d" % ((n[:84],) + r[1:]))


if __name__ == "__main__":
    main()
