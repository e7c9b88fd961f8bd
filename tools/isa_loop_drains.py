#!/usr/bin/env python3
"""List every `s_waitcnt vmcnt(0)` that sits INSIDE a loop of a kernel's gfx950 ISA — no GPU needed.

    python tools/isa_loop_drains.py [file.hip ...]        (default: every csrc/*.hip)

Round 6 found that hipcc had put a full vector-memory drain into the first K step of every step pair of the fused F(4,3) kernel (all variants,
since round 3): any load in the tile epilogue whose use sits under a lane mask made its wait-count bookkeeping give up at the loop header
("What round 6 learned").  Deliberate drains (inline asm, tile-loop epilogues, one-register-stage pipelines) show up too: read the
context before calling one a bug.  Output: kernel, [(line in the .s, loop depth)]."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-camvid_amd", "csrc")


def scan(hip):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DCVK_ABI_HASH=0", "--cuda-device-only", "-S", "-o", out, hip],
                       check=True, stderr=subprocess.DEVNULL)
        kern, depth, found = None, 0, {}
        for i, l in enumerate(open(out).read().split("\n")):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                kern, depth = m.group(1), 0
            if l.startswith(".LBB") or "; %bb." in l:
                m = re.search(r"Depth=(\d+)", l)
                depth = int(m.group(1)) if m else 0
            if kern and depth >= 1 and re.search(r"s_waitcnt vmcnt\(0\)", l) and "lgkm" not in l:
                found.setdefault(kern, []).append((i + 1, depth))
    for k, v in found.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{os.path.basename(hip)}: {name[:120]}  {v[:10]}{' ...' if len(v) > 10 else ''}")


if __name__ == "__main__":
    for f in (sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))):
        scan(f)
