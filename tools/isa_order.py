#!/usr/bin/env python3
"""Compact view of the instruction ORDER hipcc produced for the MFMA blocks of a kernel (device-only -S output):
M mfma, r ds_read, W ds_write, L buffer/global load, S store, D LDS-DMA, v VALU, a accvgpr move, | s_waitcnt, B barrier.
usage: tools/isa_order.py file.s kernel-name-substring [max chars]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]
for start in starts:
    end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
    print(lines[start].split(":")[0])
    lab = [(i, l) for i, l in enumerate(lines[start:end], start) if re.match(r"^\.LBB\d+_\d+:", l)] + [(end, "")]
    for (i, l), (j, _) in zip(lab, lab[1:]):
        n = sum(1 for x in lines[i:j] if "v_mfma" in x)
        if not n:
            continue
        seq = []
        for x in lines[i:j]:
            x = x.strip()
            if not x or x[0] in ";.":
                continue
            t = x.split()[0]
            for rx, c in ((r"v_mfma.*", "M"), (r"ds_read.*", "r"), (r"ds_write.*", "W"), (r"(buffer|global)_load_.*lds.*", "D"), (r"(buffer|global|scratch)_load.*", "L"),
                          (r"(buffer|global|scratch)_store.*", "S"), (r"v_accvgpr.*", "a"), (r"^v_.*", "v"), (r"^s_waitcnt.*", "|"), (r"^s_barrier", "B"), (r"^s_.*", "")):
                if re.match(rx, t):
                    t = c
                    break
            seq.append(t)
        print(f"  {l.split(':')[0]} lines {i}-{j}: {n} mfma")
        print("   ", "".join(seq)[:lim])
