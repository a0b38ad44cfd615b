#!/usr/bin/env python
"""Writes densematcher_amd/csrc/dm_logtab.h: the 128-entry (u_i, -log u_i) table of the fused fit's in-line logarithm."""
import math
import os
from decimal import Decimal, getcontext

getcontext().prec = 60
lines = []
for i in range(128):
    u = 1.0 / (1.0 + (i + 0.5) / 128.0)
    mant, ex = math.frexp(u)
    u = math.ldexp(round(mant * (1 << 20)) / (1 << 20), ex)       # 20 significant bits
    lines.append(f"    {u.hex()}, {float(-(Decimal(u).ln())).hex()},")
hdr = '''// Table of the in-line logarithm of dm_fitfuse.hip (generated: tools/make_logtab.py).  Entry i (the seven leading mantissa bits of the
// argument): u_i ~ 1 / (1 + (i + 1/2) / 128) with 20 significant bits, and -log(u_i) rounded from 60 decimal digits.
#pragma once
__device__ const double dm_logtab[256] = {
''' + "\n".join(lines) + "\n};\n"
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "densematcher_amd", "csrc", "dm_logtab.h"), "w").write(hdr)
