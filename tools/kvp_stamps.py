#!/usr/bin/env python
"""Decode the phase stamps of the fused dK / dV backward kernel (side build -DKVP_STAMPS=1, see attention_bwd_kvp.h):
    M4D_LIB=stamps AB_CHILD=1 python tools/ab_attn_bwd.py | grep KVPSTAMP | tail -16 > stamps.log ; python tools/kvp_stamps.py stamps.log
P-wave slots: 0 interval start, 1 elementwise step done, 2 stream done, 3 tile wait done.  dS-wave: 0 start, 1 stream done, 2 elementwise done, 3 wait done."""
import sys
rows = [l.split() for l in open(sys.argv[1]) if l.startswith("KVPSTAMP") and len(l.split()) == 12 and all(x.isdigit() for x in l.split()[3:7] + l.split()[8:12])]
for n, r in enumerate(rows):
    P = list(map(int, r[3:7])); D = list(map(int, r[8:12]))
    nxt = (int(rows[n + 1][3]) - P[0]) if n + 1 < len(rows) else 0
    print("%2d  P-wave: V %5d  M %5d  wait %4d   dS-wave: M %5d  V %5d  wait %4d   interval %5d" %
          (n, P[1] - P[0], P[2] - P[1], P[3] - P[2], D[1] - D[0], D[2] - D[1], D[3] - D[2], nxt))
