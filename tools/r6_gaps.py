#!/usr/bin/env python3
"""GPU idle time inside a step, from a rocprofv3 --kernel-trace of tools/run_leg.py:  python tools/r6_gaps.py <dir> <steps>
Every gap between the end of one kernel and the start of the next (all streams merged: a gap = NO kernel running), summed per
(kernel before -> kernel after) and divided by the steps; the first step (warm-up: JIT loads, allocations) is left out by
taking only the dispatches after the first `1/steps` of the trace's kernels."""
import collections, csv, glob, re, sys
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the leg's steps: dispatches of the product's kernels only delimit them (torch's read generation comes before)
first = next(i for i, r in enumerate(rows) if "pck::" in r[2] or "pc_spec" in r[2])
rows = rows[first:]
rows = rows[len(rows) // steps:]                    # drop the first step
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("pck::", "").replace("(anonymous namespace)::", "")[:48]
busy_end = rows[0][1]
gaps = collections.Counter(); cnt = collections.Counter()
total_gap = 0
prev = rows[0]
for r in rows[1:]:
    if r[0] > busy_end:
        g = r[0] - busy_end
        total_gap += g
        key = (short(prev[2]), short(r[2]))
        gaps[key] += g; cnt[key] += 1
    if r[1] > busy_end:
        busy_end = r[1]; prev = r
span = busy_end - rows[0][0]
n = steps - 1
print("steps measured %d: span %.2f ms / step, GPU idle %.2f ms / step (%.0f %%)" % (n, span / n / 1e6, total_gap / n / 1e6, 100.0 * total_gap / span))
for key, g in gaps.most_common(25):
    print("  %7.3f ms/step  x%-4.1f  %s  ->  %s" % (g / n / 1e6, cnt[key] / n, key[0], key[1]))
names = collections.Counter()
for r in rows:
    n = r[2]
    kind = "product" if ("pck::" in n or "pc_spec" in n) else ("copy/fill" if "rocclr" in n else "torch")
    names[kind] += 1
print("launches per step:", {k: round(v / n, 1) for k, v in names.items()} if False else {k: round(v / (steps - 1), 1) for k, v in names.items()})
tc = collections.Counter(short(r[2]) for r in rows if not ("pck::" in r[2] or "pc_spec" in r[2]))
for k, v in tc.most_common(14):
    print("  %5.1f / step  %s" % (v / (steps - 1), k))
