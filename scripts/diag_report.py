#!/usr/bin/env python3
"""Reads '[rtc-diag] ...' lines (RTC_DIAG build + RTC_DIAG_DUMP=1) and prints region time shares and, per instrumented code
section, how often a wave executed it and with how many of its 64 lanes active (the SIMT utilisation of that section)."""
import sys
REG = ["closest traversal", "container traversal", "state + pattern", "shadow traversal", "  of traversals: quirk grid scans", "lighting + spawn + pop", "  of traversals: BVH leaf tests"]
LOOPS = {0: "walk: outer iteration", 1: "walk: leaf item", 2: "program op", 3: "walk: inner node step", 4: "one-kernel: ray iteration", 5: "one-kernel: light iteration",
         8: "plane test (kernarg record)", 9: "primitive test outside a BVH", 10: "BVH leaf primitive test", 11: "quirk candidate (direction test)", 12: "quirk candidate full test",
         13: "  sphere maths", 14: "  cube maths", 15: "  cylinder / cone maths", 16: "walk set-up (frame, root)", 17: "wf: trace pass", 18: "wf: shadow pass (per light)",
         19: "wf: container pass", 20: "wf: Phong of a record", 21: "quirk grid scan", 22: "light-grid cell lookup", 23: "primitive test that reported intersections"}
rows = [list(map(int, l.split()[1:])) for l in sys.stdin if l.startswith("[rtc-diag]")]
if not rows:
    sys.exit("no [rtc-diag] lines")
d = rows[-1] + [0] * 64
total = d[14]
if not total and d[15 - 1 + 0] == 0 and d[14] == 0 and d[2 * 7]:
    # wavefront kernels: region 7 = lane-time inside work items (no whole-kernel figure)
    total = d[2 * 7]; d[15] = d[2 * 7 + 1]
if total:
    print("lanes that ran: %d, mean kernel cycles per lane: %.0f" % (d[15], total / max(1, d[15])))
    for r, name in enumerate(REG):
        if d[2 * r + 1]:
            print("  region %-24s %5.1f%% of lane-time, %d visits, %.0f cycles/visit" % (name, 100.0 * d[2 * r] / total, d[2 * r + 1], d[2 * r] / d[2 * r + 1]))
for j in sorted(LOOPS):
    it = d[16 + 2 * j + 1]
    if it:
        print("  %-34s %6.1f%% lanes active   %.3e wave executions   %.3e lane executions" % (LOOPS[j], 100.0 * d[16 + 2 * j] / (64.0 * it), it, d[16 + 2 * j]))
