#!/usr/bin/env python3
"""HBM bytes of one training step, kernel by kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (tools/profile_step_traffic.sh)
averaged per launch of each kernel symbol, times the launches of that symbol in ONE replayed step (tools/timeline.py --seq CSV).
bytes = FETCH_SIZE [KB] x 2 (gfx950 correction, MI355X_MICROARCH.md) x 1024 + WRITE_SIZE [KB] x 1024."""
import collections
import csv
import sys


def per_launch(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        k = row['Kernel_Name']
        tot[k] += float(row['Counter_Value'])
        n[k] += 1
    return {k: tot[k] / n[k] for k in tot}


def short(name):
    return name.split('(')[0].replace('void ', '').strip()[:70]


def main():
    fetch, write = per_launch(sys.argv[1], 'FETCH_SIZE'), per_launch(sys.argv[2], 'WRITE_SIZE')
    launches, dur = collections.Counter(), collections.Counter()
    for row in csv.DictReader(open(sys.argv[3])):
        launches[row['kernel']] += 1
        dur[row['kernel']] += float(row['dur_us'])
    # the sequence CSV holds truncated names: match by prefix
    def find(table, key):
        hits = [v for k, v in table.items() if short(k).startswith(key[:60]) or key.startswith(short(k)[:60])]
        return sum(hits) / len(hits) if hits else None
    rows, total, missing = [], 0.0, 0
    for k, n in launches.items():
        f, w = find(fetch, k), find(write, k)
        if f is None or w is None:
            missing += n
            continue
        b = n * (f * 2 * 1024 + w * 1024)
        rows.append((b, k, n, dur[k]))
        total += b
    rows.sort(reverse=True)
    print('HBM bytes per step (all kernels with counters): %.3f GB   (%d launches without a counter match)' % (total / 1e9, missing))
    print('%-72s %5s %10s %9s %8s' % ('kernel', 'n', 'MB/step', 'sum us', 'TB/s'))
    for b, k, n, d in rows[:45]:
        print('%-72s %5d %10.1f %9.1f %8.2f' % (k[:72], n, b / 1e6, d, b / d / 1e6 if d else 0))


if __name__ == '__main__':
    main()
