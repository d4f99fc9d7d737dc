#!/usr/bin/env python3
"""GPU occupancy of one replayed step from a rocprofv3 --kernel-trace CSV: how much of the step some kernel is running,
how much two run at once, where the idle gaps are and which kernels border the longest ones.
  python tools/timeline.py <kernel_trace.csv> [steps=10]   (the trace of `bench.py --steps N`: the last N graph replays are used)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
# the timed region: the last `steps` repetitions; find it by the largest idle gaps (fences) near the end
n = len(ev)
per = None
names = [e[2] for e in ev]
# period detection: number of kernels per step = distance between successive occurrences of the flat Adam kernel
adam = [i for i, nm in enumerate(names) if 'flat_adam' in nm]
if len(adam) >= steps + 1:
    lo, hi = adam[-steps - 1] + 1, adam[-1] + 1
else:                                       # inference: the timed replays are the longest gap-free stretch of the trace
    cuts = [0]
    run_end = ev[0][1]
    for i in range(1, n):
        if ev[i][0] - run_end > 150000:     # 150 us of idle GPU: a host synchronisation
            cuts.append(i)
        run_end = max(run_end, ev[i][1])
    cuts.append(n)
    lo, hi = max(((cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)), key=lambda ab: ab[1] - ab[0])
seg = ev[lo:hi]
t0, t1 = seg[0][0], max(e[1] for e in seg)
span = t1 - t0
# sweep
pts = []
for s, e, _ in seg:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy1 = busy2 = 0
depth = 0
last = t0
gaps = []
for t, d in pts:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    if depth == 0 and t > last: gaps.append((t - last, last))
    depth += d
    last = t
print('kernels in window: %d (%.1f per step), window %.3f ms per step' % (len(seg), len(seg) / steps, span / steps / 1e6))
print('some kernel running: %.1f %%   two or more: %.1f %%   idle: %.1f %% (%.3f ms per step)' % (
    100.0 * busy1 / span, 100.0 * busy2 / span, 100.0 * (span - busy1) / span, (span - busy1) / steps / 1e6))
ksum = sum(e[1] - e[0] for e in seg)
print('sum of kernel durations: %.3f ms per step' % (ksum / steps / 1e6))
gaps.sort(reverse=True)
hist = {}
for g, _ in gaps:
    b = 1 if g < 2000 else 2 if g < 5000 else 5 if g < 10000 else 10 if g < 50000 else 50
    hist[b] = hist.get(b, [0, 0]); hist[b][0] += 1; hist[b][1] += g
for b in sorted(hist):
    print('  gaps %s us: %5d per step, %.3f ms per step' % ({1: '<2', 2: '2-5', 5: '5-10', 10: '10-50', 50: '>50'}[b], hist[b][0] / steps, hist[b][1] / steps / 1e6))
ends = {e[1]: e[2] for e in seg}
starts = {e[0]: e[2] for e in seg}
print('longest gaps:')
for g, at in gaps[:12]:
    before = ends.get(at, '?')
    after = starts.get(at + g, '?')
    print('  %7.1f us after %-60s before %s' % (g / 1e3, before[:60], after[:60]))
if len(sys.argv) > 3:                       # per-step launch counts by kernel name
    cnt = {}
    for s, e, nm in seg:
        c = cnt.setdefault(nm, [0, 0]); c[0] += 1; c[1] += e - s
    print('launches per step by kernel:')
    for nm, (c, d) in sorted(cnt.items(), key=lambda kv: -kv[1][1]):
        print('  %6.1f x %8.1f us  %s' % (c / steps, d / c / 1e3, nm[:110]))
