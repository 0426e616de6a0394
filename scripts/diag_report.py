#!/usr/bin/env python3
"""Reads '[rtc-diag] ...' lines (RTC_DIAG build + RTC_DIAG_DUMP=1) and prints region time shares and loop lane utilisation."""
import sys
REG = ["closest traversal", "container traversal", "state + pattern", "shadow traversal", "  of traversals: quirk grid scans", "lighting + spawn + pop", "  of traversals: BVH leaf tests"]
LOOPS = ["bvh_walk iteration", "leaf item", "program op", "inner node", "ray iteration", "light iteration"]
rows = [list(map(int, l.split()[1:])) for l in sys.stdin if l.startswith("[rtc-diag]")]
if not rows:
    sys.exit("no [rtc-diag] lines")
d = rows[-1]
total = d[14]
print("lanes that ran: %d, mean kernel cycles per lane: %.0f" % (d[15], total / max(1, d[15])))
for r, name in enumerate(REG):
    if d[2 * r + 1]:
        print("  region %-24s %5.1f%% of lane-time, %d visits, %.0f cycles/visit" % (name, 100.0 * d[2 * r] / total, d[2 * r + 1], d[2 * r] / d[2 * r + 1]))
for j, name in enumerate(LOOPS):
    it = d[16 + 2 * j + 1]
    if it:
        print("  loop   %-24s %6.1f%% lanes active (%.2e wave-iterations)" % (name, 100.0 * d[16 + 2 * j] / (64.0 * it), it))
