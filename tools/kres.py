#!/usr/bin/env python3
"""Developer tool: per-function resource summary (registers, scratch bytes per lane, code size, scratch loads / stores) of a region-stage
assembly listing (tools/kres.sh writes /tmp/kres_w4.s and /tmp/kres_w8.s).    tools/kres.py /tmp/kres_w4.s"""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split("\n")
start = 0
for i, l in enumerate(lines):
    m = re.match(r"\t\.size\t(\S+), \.Lfunc_end", l)
    if not m:
        continue
    body = lines[start:i]
    info = "\n".join(lines[i:i + 25])
    start = i
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    g = lambda k: (re.search(k + r"[:=] *(\d+)", info) or [0, "?"])[1]
    st = sum(1 for x in body if re.match(r"\s*scratch_store", x)); ld = sum(1 for x in body if re.match(r"\s*scratch_load", x))
    print("%-40s vgprs %3s sgprs %3s scratch %4s B code %6s B  scratch stores %3d loads %3d" % (name[-40:], g("NumVgprs"), g("NumSgprs"), g("ScratchSize"), g("codeLenInByte "), st, ld))
