#!/usr/bin/env python3
"""Critical path of one replayed training step from its launch-by-launch sequence (profiles/rNN/train_step_sequence.csv, written by
tools/timeline.py --seq: start_us, dur_us, queue, kernel).

Walk back from the kernel that ends last; a kernel's blocker is the LATEST-ENDING predecessor among (a) the previous launch of its
own queue and (b) the last launch of the other queue that ended before it started — a heuristic (the trace carries no dependency
edges: a cross-queue edge is assumed wherever the other queue's kernel ended later than the own queue's predecessor), good enough to
say which kernel families the wall time hangs on.

    python tools/critical_path.py profiles/r05/train_step_sequence.csv [--top 40]
"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--list', default=None, help='"a:b" (microseconds): print the launches of the path that start inside the window')
    a = ap.parse_args()
    K = sorted((float(r['start_us']), float(r['start_us']) + float(r['dur_us']), r['queue'], r['kernel'][:60]) for r in csv.DictReader(open(a.csv)))
    byq = collections.defaultdict(list)
    for i, k in enumerate(K):
        byq[k[2]].append(i)
    prev = {}
    for l in byq.values():
        for x, y in zip(l, l[1:]):
            prev[y] = x
    i = max(range(len(K)), key=lambda j: K[j][1])
    path = []
    while i is not None:
        path.append(i)
        s, q = K[i][0], K[i][2]
        cand = [(K[prev[i]][1], prev[i])] if i in prev else []
        for oq, l in byq.items():
            if oq != q:
                before = [j for j in l if K[j][1] <= s + 1.0]
                if before:
                    j = max(before, key=lambda j: K[j][1])
                    cand.append((K[j][1], j))
        i = max(cand)[1] if cand else None
    path.reverse()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for i in path:
        e = agg[K[i][3]]
        e[0] += 1
        e[1] += K[i][1] - K[i][0]
    tot = sum(K[i][1] - K[i][0] for i in path)
    gaps = sum(max(0.0, K[b][0] - K[x][1]) for x, b in zip(path, path[1:]))
    sw = sum(1 for x, b in zip(path, path[1:]) if K[x][2] != K[b][2])
    onq = collections.Counter(K[i][2] for i in path)
    print('launches in the step %d, on the critical path %d (%s); kernel time on the path %.1f us + gaps %.1f us; queue changes %d'
          % (len(K), len(path), ', '.join('%s: %d' % kv for kv in sorted(onq.items())), tot, gaps, sw))
    print('%-60s %5s %10s' % ('kernel', 'n', 'us on path'))
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print('%-60s %5d %10.1f' % (k, n, us))
    if a.list:
        lo, hi = (float(v) for v in a.list.split(':'))
        for i in path:
            if lo <= K[i][0] <= hi:
                print('%10.1f %8.1f %s %s' % (K[i][0], K[i][1] - K[i][0], K[i][2], K[i][3]))


if __name__ == '__main__':
    main()
