#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one per pass) for one kernel-name substring:
   python tools/pmc_summary.py <dir> <kernel substring>  -> averages per launch as JSON."""
import csv, glob, json, os, sys
d, sub = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(os.path.join(d, '*.csv'))):
    acc, n = {}, {}
    for row in csv.DictReader(open(f)):
        if sub not in row.get('Kernel_Name', ''):
            continue
        c, v = row['Counter_Name'], float(row['Counter_Value'])
        acc[c] = acc.get(c, 0.0) + v
        n[c] = n.get(c, 0) + 1
    for c in acc:
        out[c] = acc[c] / n[c]
w = out.get('SQ_WAVES', 0)
if w:
    out['per_wave'] = {k[9:].lower(): round(out[k] / w, 1) for k in out if k.startswith('SQ_INSTS_')}
    wc = out.get('SQ_WAVE_CYCLES', 0)
    if wc:
        out['frac_of_wave_cycles'] = {k: round(out[k] / wc, 3) for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if k in out}
if 'FETCH_SIZE' in out:
    out['hbm_read_MB_corrected_x2'] = round(out['FETCH_SIZE'] * 2 / 1024, 2)      # KB units; gfx950 x2 (MI355X_MICROARCH.md)
if 'WRITE_SIZE' in out:
    out['hbm_write_MB'] = round(out['WRITE_SIZE'] / 1024, 2)
print(json.dumps(out, indent=1))
