#!/usr/bin/env python3
"""Copies the outputs of tools/profile_bench.sh (rocprofv3 --kernel-trace --stats over `bench.py`, merged back under
gpurun_out/<src>/) into profiles/<round>/final_* and prints the live-vs-rocprof agreement table for its README.
    python tools/refresh_profiles.py [src dir under gpurun_out = prof4] [round dir under profiles = r01]"""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
SRC = os.path.join(ROOT, 'gpurun_out', sys.argv[1] if len(sys.argv) > 1 else 'prof4')
DST = os.path.join(ROOT, 'profiles', sys.argv[2] if len(sys.argv) > 2 else 'r01')
os.makedirs(DST, exist_ok=True)


def main():
    print('| mode | kernel | live avg us | rocprof avg us | launches/step |\n|---|---|---|---|---|')
    for mode in ('train', 'fwd'):
        rows = list(csv.DictReader(open(os.path.join(SRC, '%s_kernel_stats.csv' % mode))))
        with open(os.path.join(DST, 'final_bf16_%s_kernel_stats_top40.csv' % mode), 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            for r in rows[:40]:
                w.writerow(r)
        shutil.copy(os.path.join(SRC, 'bench_%s.json' % mode), os.path.join(DST, 'final_bf16_%s_bench.json' % mode))
        bench = json.load(open(os.path.join(SRC, 'bench_%s.json' % mode)))
        prof = {}
        for r in rows:
            name = r['Name'].replace('void ', '').split('(')[0]
            prof[name] = float(r['AverageNs']) / 1e3
        ks = [(k, v) for k, v in bench['kernels'].items() if k.startswith('mpconv_')]
        for k, v in sorted(ks, key=lambda kv: -kv[1]['total_ms']):
            if k in prof:
                print('| %s | `%s` | %.1f | %.1f | %d |' % (mode, k, v['avg_us'], prof[k], v['launches']))
        print('<!-- %s: %.2f ms/step under the profiler -->' % (mode, bench['ms_per_step']))


if __name__ == '__main__':
    main()
