#!/usr/bin/env python3
"""Registers, spills, LDS and scratch of every kernel in libpptoas_hip.so, read from
the code object's metadata (no GPU needed):  python tools/kernel_resources.py [filter]

... and, from the disassembly, WHERE the scratch traffic of a kernel sits (loop_scratch): the metadata's
"spilled VGPRs" cannot tell the saves and restores around the one call to tail_work (executed once per
ticket a wave draws) from spills inside the row loop (executed per row, and a scratch load queues behind
the prefetched row like every other vector-memory access).  A loop is a backward branch; a scratch
instruction, or an SGPR spill to a VGPR lane (v_writelane / v_readlane), counts as "in a loop" when
some loop that contains NO call (s_swappc) spans it -- the phase loop around the call site is not one.
    python tools/kernel_resources.py --loops [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    return co


def loop_scratch(so=None, pat=""):
    """{kernel (demangled, short): dict(scratch_total, scratch_in_loops, lane_spills_in_loops, calls, loops)} for the
    kernels whose name contains `pat`."""
    so = so or os.path.join(ROOT, "pulseportraiture_amd", "csrc", "libpptoas_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(so, tmp)
        full = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True,
                              check=True).stdout
        blocks, cur = {}, None
        for ln in full.splitlines():
            m = re.match(r"^([0-9a-f]+) <([^>]+)>:", ln)
            if m:
                cur = m.group(2)
                blocks[cur] = [ln]
            elif cur is not None:
                blocks[cur].append(ln)
        want = [k for k in blocks if k.startswith("_ZN2pp")]
        dem = subprocess.run(["c++filt"], input="\n".join(want), capture_output=True, text=True).stdout.split("\n")
        out = {}
        for sym, name in zip(want, dem):
            short = re.sub(r"\(.*\)$", "", re.sub(r"^(void )?pp::", "", name))
            if pat and pat not in short:
                continue
            if not short.startswith("k_"):
                continue
            dis = "\n".join(blocks[sym])
            ins = []          # (address, mnemonic, branch target or None)
            start = None
            for ln in dis.splitlines():
                m = re.match(r"^([0-9a-f]+) <", ln)
                if m:
                    start = int(m.group(1), 16)
                    continue
                m = re.match(r"^\s+(\S+).*//\s*([0-9A-Fa-f]+):", ln)
                if not m or start is None:
                    continue
                mnem, addr = m.group(1), int(m.group(2), 16)
                tgt = None
                if mnem.startswith("s_cbranch") or mnem == "s_branch":
                    t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", ln)
                    if t:
                        tgt = start + int(t.group(1), 16)
                    elif re.search(r"<[^>+]*>", ln):
                        tgt = start
                ins.append((addr, mnem, tgt))
            loops = [(t, a) for a, mn, t in ins if t is not None and t <= a]
            calls = [a for a, mn, t in ins if mn.startswith("s_swappc")]
            callfree = [(lo, hi) for lo, hi in loops if not any(lo <= cc <= hi for cc in calls)]
            inloop = lambda a: any(lo <= a <= hi for lo, hi in callfree)
            scr = [a for a, mn, t in ins if mn.startswith("scratch_")]
            lane = [a for a, mn, t in ins if mn.startswith("v_writelane") or mn.startswith("v_readlane")]
            out[short] = dict(scratch_total=len(scr), scratch_in_loops=sum(1 for a in scr if inloop(a)),
                              lane_spills_total=len(lane), lane_spills_in_loops=sum(1 for a in lane if inloop(a)),
                              calls=len(calls), loops=len(loops), instructions=len(ins))
        return out


def main():
    so = os.path.join(ROOT, "pulseportraiture_amd", "csrc", "libpptoas_hip.so")
    if len(sys.argv) > 1 and sys.argv[1] == "--loops":
        res = loop_scratch(so, sys.argv[2] if len(sys.argv) > 2 else "")
        print("%-60s %8s %8s %8s %8s %6s %6s" % ("kernel", "scratch", "in-loop", "lanespl", "in-loop", "calls", "loops"))
        for k, v in sorted(res.items()):
            print("%-60s %8d %8d %8d %8d %6d %6d" % (k[:60], v["scratch_total"], v["scratch_in_loops"], v["lane_spills_total"],
                                                      v["lane_spills_in_loops"], v["calls"], v["loops"]))
        return
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
