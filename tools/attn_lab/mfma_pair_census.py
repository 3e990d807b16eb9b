#!/usr/bin/env python
"""Static census of the instruction pattern the round-2 attention incident was bisected to: two
`v_mfma_f32_32x32x16_bf16` that are adjacent in the instruction stream and share srcA.  For every such pair: the VGPR
bank alignment (first register index mod 4) of srcA and of the two srcB tuples.

The failing build has srcA and both srcB tuples 4-aligned (0, 0, 0) in 20 of its 25 adjacent pairs; its clean sibling
(`r2ship`) has srcB at 4k + 2 -- which made operand-bank conflicts the suspect.  The census of the SHIPPED kernel says that
is not sufficient: 26 of its 28 adjacent pairs are (0, 0, 0) as well, and it reproduces its output bit for bit under every
timing perturbation of the stress suite.  (No GPU needed: reads device assembly listings.)

usage: python tools/attn_lab/mfma_pair_census.py [listing.s ...]   (default: profiles/r03a_isa/*.s and the build's _isa cache)"""
import collections
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def census(path):
    lines = open(path).read().split("\n")
    mf = [(i, l) for i, l in enumerate(lines) if re.match(r"\s+v_mfma_f32_32x32x16_bf16", l)]
    out = []
    for (i, a), (j, b) in zip(mf, mf[1:]):
        between = [x for x in lines[i + 1:j] if re.match(r"\s+[a-z_]", x) and not x.strip().startswith(";")]
        ra, rb = re.findall(r"v\[(\d+):(\d+)\]", a), re.findall(r"v\[(\d+):(\d+)\]", b)
        if len(ra) >= 3 and len(rb) >= 3 and ra[1] == rb[1]:
            out.append((len(between), int(ra[1][0]) % 4, int(ra[2][0]) % 4, int(rb[2][0]) % 4))
    return out


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r03a_isa", "*.s"))) + \
        [os.path.join(ROOT, "hi3d-official_amd", "hi3d_hip", "_isa", "attention.hip.s")]
    for f in files:
        if not os.path.exists(f):
            continue
        r = census(f)
        adj = [x for x in r if x[0] == 0]
        c = collections.Counter((x[1], x[2], x[3]) for x in adj)
        print(f"{os.path.basename(f):32s} srcA-sharing consecutive MFMA pairs {len(r):3d}, directly adjacent {len(adj):3d}; "
              f"(srcA, srcB1, srcB2) mod 4: {dict(c.most_common())}")
