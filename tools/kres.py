#!/usr/bin/env python3
"""tools/kres.py <asm.s> [name-substring...]: register / LDS / spill figures of the kernels in a gfx950 assembly listing
(hipcc -S --cuda-device-only), one line per kernel.  Developer aid."""
import re, sys
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pats and not any(p in name for p in pats):
        continue
    print(f"vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>4} sgpr {g('sgpr_count'):>4} vspill {g('vgpr_spill_count'):>4} "
          f"sspill {g('sgpr_spill_count'):>4} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}  {name[:110]}")
