#!/usr/bin/env python3
"""profiles/rNN/pmc_infer.json from the five counter_collection CSVs tools/profile_pmc_infer.sh leaves:
   python tools/pmc_infer_summary.py gpurun_out/r06/pmc_infer > profiles/r06/pmc_infer.json"""
import csv, glob, json, os, re, sys
d = sys.argv[1]
KERNELS = ('factor_layer_fwd_kernel', 'mpconv_block_fwd_kernel', 'mpconv_block_fanin_kernel', 'mpconv_block_fanout_kernel',
           'mpconv_block_rows1_kernel', 'linear_instnorm_fwd_kernel')
acc = {}
for f in sorted(glob.glob(os.path.join(d, '*.csv'))):
    for row in csv.DictReader(open(f)):
        name = row.get('Kernel_Name', '')
        if not any(k in name for k in KERNELS):
            continue
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*$', '', name)
        a = acc.setdefault(name, {})
        c = row['Counter_Name']
        s, n = a.get(c, (0.0, 0))
        a[c] = (s + float(row['Counter_Value']), n + 1)
out = {'_doc': 'tools/profile_pmc_infer.sh: five separate rocprofv3 --kernel-trace --pmc passes over `python bench.py --mode fwd` (B = 4096, bf16); '
               'averages per launch of each one-kernel-per-block / per-layer kernel (tools/pmc_infer_summary.py); HBM bytes = FETCH_SIZE x 2 (gfx950) + WRITE_SIZE',
       'kernels': {}}
for name in sorted(acc):
    v = {c: s / n for c, (s, n) in acc[name].items()}
    k = {}
    if 'FETCH_SIZE' in v:
        k['hbm_read_MB_corrected_x2'] = round(v['FETCH_SIZE'] * 2 / 1024, 2)
    if 'WRITE_SIZE' in v:
        k['hbm_write_MB'] = round(v['WRITE_SIZE'] / 1024, 2)
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        k['traffic_bytes_per_launch'] = int((v['FETCH_SIZE'] * 2 + v['WRITE_SIZE']) * 1024)
    if v.get('SQ_BUSY_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        k['mfma_busy'] = round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * v['SQ_BUSY_CYCLES']), 4)      # (the definition of tools/refresh_profiles_r05.py)
    if v.get('SQ_LDS_IDX_ACTIVE'):
        k['lds_bank_conflict_frac'] = round(v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE'], 4)
    w = v.get('SQ_WAVES', 0)
    if w:
        k['per_wave'] = {c[9:].lower(): round(v[c] / w, 1) for c in sorted(v) if c.startswith('SQ_INSTS_')}
        wc = v.get('SQ_WAVE_CYCLES', 0)
        if wc:
            k['frac_of_wave_cycles'] = {c: round(v[c] / wc, 3) for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if c in v}
    out['kernels'][name] = k
print(json.dumps(out, indent=1))
