#!/usr/bin/env python
"""From gpurun_out/timeline.txt of a stamps build (slot, start, end, children-there, front, parent front; 10 ns ticks): per tree
level the front's own time after its children arrived (end - children there) and the hand-off (children there - end of the later child)."""
import sys
import numpy as np
t = np.loadtxt(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline.txt")
t = t[t[:, 1] > 0]
base = t[:, 1].min()
end_of = {int(r[4]): (r[2] - base) * 0.01 for r in t}
kids = {}
for r in t:
    kids.setdefault(int(r[5]), []).append((r[2] - base) * 0.01)
a, n = 0, (len(t) + 1) // 2
while n >= 1 and a < len(t):
    own, hand, spread = [], [], []
    for r in t[a:a + n]:
        f = int(r[4]); there = (r[3] - base) * 0.01; end = (r[2] - base) * 0.01
        own.append(end - there)
        if f in kids:
            hand.append(there - max(kids[f]))
            if len(kids[f]) > 1: spread.append(max(kids[f]) - min(kids[f]))
    fmt = lambda v: "%6.2f %6.2f %6.2f" % (min(v), float(np.median(v)), max(v)) if v else "     -      -      -"
    print("level n %4d  own min/med/max %s   hand-off %s   sibling spread %s" % (n, fmt(own), fmt(hand), fmt(spread)))
    a += n; n //= 2
