#!/bin/sh
# Run on the GPU box: SQ issue counters + FETCH / WRITE passes of the synthetic-PGM operator's backward (tools/xbench.py, batch 1024).
#   sh tools/profile_pmc_ext.sh <outdir under gpurun_out>         (each counter set is its own rocprofv3 run with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
run() { # name, counters
    rm -rf /tmp/px_$1
    timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/px_$1 -o $1 -- python tools/xbench.py 1024 > /tmp/px_$1.log 2>&1
    find /tmp/px_$1 -name "*counter_collection.csv" -exec cp {} $OUT/$1.csv \;
}
run issue_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run issue_b "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run issue_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_MOPS_BF16"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python - $OUT <<'PY'
import csv, glob, json, os, sys
d = sys.argv[1]
out = {}
for f in sorted(glob.glob(os.path.join(d, '*.csv'))):
    acc, n = {}, {}
    for row in csv.DictReader(open(f)):
        kn = row.get('Kernel_Name', '')
        if 'mpconv_bwd_ext' not in kn:
            continue
        key = kn.split('(')[0].replace('void ', '')
        c, v = row['Counter_Name'], float(row['Counter_Value'])
        acc[(key, c)] = acc.get((key, c), 0.0) + v
        n[(key, c)] = n.get((key, c), 0) + 1
    for (key, c) in acc:
        out.setdefault(key, {})[c] = acc[(key, c)] / n[(key, c)]
for key, o in out.items():
    w = o.get('SQ_WAVES', 0)
    if w:
        o['per_wave'] = {k[9:].lower(): round(o[k] / w, 1) for k in list(o) if k.startswith('SQ_INSTS_')}
        wc = o.get('SQ_WAVE_CYCLES', 0)
        if wc:
            o['frac_of_wave_cycles'] = {k: round(o[k] / wc, 3) for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS') if k in o}
    if 'FETCH_SIZE' in o and 'WRITE_SIZE' in o:
        o['traffic_bytes_per_launch'] = int(o['FETCH_SIZE'] * 2 * 1024 + o['WRITE_SIZE'] * 1024)      # KB units; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md)
    if o.get('SQ_BUSY_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in o:
        o['mfma_busy'] = round(o['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * o['SQ_BUSY_CYCLES']), 4)
    if o.get('SQ_LDS_IDX_ACTIVE'):
        o['lds_bank_conflict_frac'] = round(o.get('SQ_LDS_BANK_CONFLICT', 0.0) / o['SQ_LDS_IDX_ACTIVE'], 4)
json.dump(out, open(os.path.join(d, 'pmc_ext_bwd.json'), 'w'), indent=1, sort_keys=True)
for key, o in out.items():
    print(key, {k: o[k] for k in ('per_wave', 'frac_of_wave_cycles', 'mfma_busy', 'lds_bank_conflict_frac', 'traffic_bytes_per_launch') if k in o})
PY
