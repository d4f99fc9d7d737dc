"""Wall-time attribution of one replayed training step from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

    python tools/timeline.py <kernel_trace.csv> [--marker flat_adam] [--top 40]

The step is the window between the ends of the last two launches of the marker kernel (one per step).  Inside it every
instant is attributed to the kernels running then, 1/k each when k run concurrently (two streams), and idle instants to the
kernel that starts next ("gap before").  Sum of the attributed times = the wall time of the step."""
import argparse
import collections
import csv
import json
import re


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--marker', default='flat_adam')
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--marker-stride', type=int, default=1, help='launches of the marker kernel per step')
    ap.add_argument('--marker-skip', type=int, default=0, help='ignore the last N launches of the marker kernel (e.g. eager steps behind the replayed ones)')
    ap.add_argument('--json', default=None)
    ap.add_argument('--dump', default=None, help='kernel-name substring: list, per queue, what runs between consecutive launches of it')
    ap.add_argument('--dump-from', type=int, default=0)
    ap.add_argument('--dump-count', type=int, default=2)
    ap.add_argument('--after', default=None, help='kernel-name substring: the window starts at the END of its last launch inside the step (a head-start spin kernel)')
    ap.add_argument('--seq', default=None, help='write the step window launch by launch (start offset us, duration us, queue, kernel) as CSV')
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
    rows.sort()
    marks = [r for r in rows if a.marker in r[2]]
    assert len(marks) >= 2, 'marker kernel not found twice'
    if a.marker_skip:
        marks = marks[:-a.marker_skip]
    assert len(marks) > a.marker_stride
    t0, t1 = marks[-1 - a.marker_stride][1], marks[-1][1]
    win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    if a.after:
        heads = [r for r in win if a.after in r[2]]
        if heads:
            t0 = max(r[1] for r in heads)
            win = [r for r in win if r[0] >= t0]
    ev = []
    for i, (s, e, _, _, _) in enumerate(win):
        ev.append((s, 1, i)), ev.append((e, 0, i))
    ev.sort()
    live, last = set(), t0
    attr = collections.Counter()
    gap_before = collections.Counter()
    conc = collections.Counter()
    idle_from = None
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            conc[min(len(live), 3)] += dt
            if live:
                for j in live:
                    attr[short(win[j][2])] += dt / len(live)
            else:
                idle_from = (idle_from or 0) + dt
        if kind == 1:
            if not live and idle_from:
                gap_before[short(win[i][2])] += idle_from
                idle_from = None
            live.add(i)
        else:
            live.discard(i)
        last = t
    wall = t1 - t0
    dur = collections.Counter()
    cnt = collections.Counter()
    for s, e, n, _, _ in win:
        dur[short(n)] += e - s
        cnt[short(n)] += 1
    queues = collections.Counter()
    for s, e, n, q, st in win:
        queues[(q, st)] += e - s
    out = {'wall_ms': wall / 1e6, 'launches': len(win), 'sum_kernel_ms': sum(dur.values()) / 1e6,
           'idle_ms': conc[0] / 1e6, 'one_kernel_ms': conc[1] / 1e6, 'two_kernels_ms': conc[2] / 1e6, 'three_plus_ms': conc[3] / 1e6,
           'queues_busy_ms': {'%s/%s' % k: v / 1e6 for k, v in queues.items()}}
    print(json.dumps(out))
    print('%-72s %5s %9s %9s %9s' % ('kernel', 'n', 'attr_ms', 'sum_ms', 'gap_ms'))
    for n, v in attr.most_common(a.top):
        print('%-72s %5d %9.3f %9.3f %9.3f' % (n, cnt[n], v / 1e6, dur[n] / 1e6, gap_before[n] / 1e6))
    print('total attributed %.3f ms + idle %.3f ms = wall %.3f ms' % (sum(attr.values()) / 1e6, conc[0] / 1e6, wall / 1e6))
    # cross-stream waits: for every queue, the time it sits idle while ANOTHER queue runs a kernel, charged to the kernel it starts
    # next (the consumer of the join)
    waits = {}
    for q in sorted(set((r[3], r[4]) for r in win)):
        mine = [r for r in win if (r[3], r[4]) == q]
        other = sorted((r[0], r[1]) for r in win if (r[3], r[4]) != q)
        by, tot, last, oi = collections.Counter(), 0, t0, 0
        for s_, e_, n_, _, _ in mine:
            if s_ > last:
                busy = 0
                for os_, oe_ in other:
                    if oe_ <= last:
                        continue
                    if os_ >= s_:
                        break
                    busy += min(oe_, s_) - max(os_, last)
                by[short(n_)] += busy
                tot += busy
            last = max(last, e_)
        waits['%s/%s' % q] = {'kernels': len(mine), 'waiting_while_other_busy_ms': tot / 1e6,
                              'resumes_with': {n: v / 1e6 for n, v in by.most_common(10)}}
        print('queue %s/%s: %d kernels, idle while the other queue is busy %.3f ms; it resumes with:' % (q[0], q[1], len(mine), tot / 1e6))
        for n, v in by.most_common(10):
            print('    %-70s %.3f ms' % (n, v / 1e6))
    out['cross_stream_waits'] = waits
    if a.dump:
        hits = [r for r in win if a.dump in r[2]]
        for k in range(a.dump_from, min(a.dump_from + a.dump_count, len(hits) - 1)):
            w0, w1 = hits[k][0], hits[k + 1][1]
            print('--- window %d: %.1f us, from %s to the next one' % (k, (w1 - w0) / 1e3, short(hits[k][2])))
            for q in sorted(set((r[3], r[4]) for r in win)):
                print('  queue %s/%s' % q)
                last = w0
                for s_, e_, n_, q3, q4 in win:
                    if (q3, q4) != q or e_ < w0 or s_ > w1:
                        continue
                    print('    +%7.1f  %6.1f us  %s%s' % ((s_ - w0) / 1e3, (e_ - s_) / 1e3, short(n_), ('   [gap %.1f]' % ((s_ - last) / 1e3)) if s_ - last > 4000 else ''))
                    last = e_
    if a.seq:
        with open(a.seq, 'w') as f:
            f.write('start_us,dur_us,queue,kernel\n')
            for s_, e_, n_, q3, q4 in win:
                f.write('%.2f,%.2f,%s/%s,"%s"\n' % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, q3, q4, short(n_)))
    if a.json:
        out['kernels'] = {n: {'n': cnt[n], 'attr_ms': attr[n] / 1e6, 'sum_ms': dur[n] / 1e6, 'gap_before_ms': gap_before[n] / 1e6} for n in dur}
        json.dump(out, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
