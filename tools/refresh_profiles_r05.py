#!/usr/bin/env python3
"""profiles/r05/ from gpurun_out/r05/ (the output of tools/profile_r05.sh on the GPU box):
  * final_bf16_{train,fwd}_kernel_stats_top40.csv + the bench line printed inside the profiled run,
  * pmc_<instance>.json: SQ issue counters + FETCH_SIZE / WRITE_SIZE per launch of each parity-kernel INSTANCE (V->F and
    F->V measured separately) and of the fused block-tail kernels,
  * pmc_traffic.json: what bench.py's roofline object reads — HBM bytes per launch (FETCH_SIZE x 2 for gfx950 + WRITE_SIZE,
    MI355X_MICROARCH.md) and the matrix-core busy fraction, keyed by kernel symbol,
  * bench_*.json: bench.py lines of the final code and of the A/B switches of the round's levers; kbench_* / lbench / wbench /
    tbench / sbench / mbench: stand-alone device times; cpu_baseline_all_cores.json; train_step_{timeline,sequence,traffic}."""
import csv
import glob
import json
import os
import shutil

import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r05'      # python tools/refresh_profiles_r05.py r06: the same layout for a later round
SRC = os.path.join(ROOT, 'gpurun_out', ROUND)
DST = os.path.join(ROOT, 'profiles', ROUND)
os.makedirs(DST, exist_ok=True)


def summarise(d, sub):
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
        out['hbm_read_MB_corrected_x2'] = round(out['FETCH_SIZE'] * 2 / 1024, 2)      # KB units; gfx950 x 2
    if 'WRITE_SIZE' in out:
        out['hbm_write_MB'] = round(out['WRITE_SIZE'] / 1024, 2)
    if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
        out['traffic_bytes_per_launch'] = int(out['FETCH_SIZE'] * 2 * 1024 + out['WRITE_SIZE'] * 1024)
    if out.get('SQ_BUSY_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in out:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4), each 8 CUs = 32 SIMDs; MFMA_BUSY over all SIMDs
        out['mfma_busy'] = round(out['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * out['SQ_BUSY_CYCLES']), 4)
    if out.get('SQ_LDS_IDX_ACTIVE'):
        out['lds_bank_conflict_frac'] = round(out.get('SQ_LDS_BANK_CONFLICT', 0.0) / out['SQ_LDS_IDX_ACTIVE'], 4)
    return out


def kernel_names(d):
    names = set()
    for f in glob.glob(os.path.join(d, '*.csv')):
        for row in csv.DictReader(open(f)):
            names.add(row.get('Kernel_Name', ''))
    return names


def main():
    # kernel statistics
    for mode in ('train', 'fwd', 'train1'):       # train1: the training step on ONE stream (tools/profile_bench.sh)
        src = os.path.join(SRC, 'prof', '%s_kernel_stats.csv' % mode)
        if not os.path.exists(src):
            continue
        rows = list(csv.DictReader(open(src)))
        with open(os.path.join(DST, 'final_bf16_%s_kernel_stats_top40.csv' % mode), 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            for r in rows[:40]:
                w.writerow(r)
        shutil.copy(os.path.join(SRC, 'prof', 'bench_%s.json' % mode), os.path.join(DST, 'final_bf16_%s_bench.json' % mode))
    traffic = {'_doc': 'Per launch, from separate rocprofv3 --kernel-trace --pmc passes (tools/profile_r05.sh -> tools/profile_pmc_fwd.sh / '
                       'profile_pmc_tail.sh) at B = 4096 codewords: traffic_bytes_per_launch = FETCH_SIZE (KB) x 2 (gfx950 correction, '
                       'MI355X_MICROARCH.md) x 1024 + WRITE_SIZE (KB) x 1024; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) '
                       '(SQ_BUSY_CYCLES counts per shader engine, 32 SIMDs each).  Every parity-kernel instance has its OWN passes.',
               'kernels': {}}
    inst = [('pmc_fwd_v2f', 'mpconv_fwd_ws_kernel'), ('pmc_fwd_f2v', 'mpconv_fwd_ws_kernel'), ('pmc_bwd_v2f', 'mpconv_bwd_ws_kernel'),
            ('pmc_bwd_f2v', 'mpconv_bwd_ws_kernel')]
    for d, sub in inst:
        dd = os.path.join(SRC, d)
        if not os.path.isdir(dd):
            continue
        s = summarise(dd, sub)
        json.dump(s, open(os.path.join(DST, d + '.json'), 'w'), indent=1)
        for f in ('fetch.csv', 'write.csv'):
            if os.path.exists(os.path.join(dd, f)):
                shutil.copy(os.path.join(dd, f), os.path.join(DST, '%s_%s' % (d, f)))
        for nm in kernel_names(dd):
            if sub in nm:
                sym = nm.replace('void ', '').split('(')[0]
                traffic['kernels'][sym] = {k: s[k] for k in ('traffic_bytes_per_launch', 'mfma_busy', 'lds_bank_conflict_frac') if k in s}
    dd = os.path.join(SRC, 'pmc_tail')
    if os.path.isdir(dd):
        for nm in sorted(kernel_names(dd)):
            if 'block_tail_kernel' in nm:
                sym = nm.replace('void ', '').split('(')[0]
                s = summarise(dd, nm)
                json.dump(s, open(os.path.join(DST, 'pmc_tail_%s.json' % sym.replace('block_tail_kernel', 'mode').replace('<', '').replace('>', '')
                                               .replace(', ', '_')), 'w'), indent=1)
                traffic['kernels'][sym] = {k: s[k] for k in ('traffic_bytes_per_launch', 'mfma_busy') if k in s}
    json.dump(traffic, open(os.path.join(DST, 'pmc_traffic.json'), 'w'), indent=1)
    import glob as _g
    names = [os.path.basename(f) for f in _g.glob(os.path.join(SRC, 'bench_*.json'))] + ['cpu_baseline_all_cores.json', 'lbench.log',
                                                                                       'wbench.log', 'tbench.log', 'sbench.log', 'mbench.log', 'wmbench.log',
                                                                                       'wbench_register_direct.log', 'wmbench_register_direct.log', 'fastpath_step.log']
    dd = os.path.join(SRC, 'pmc_wgrad')          # round 6: HBM bytes per launch of the weight-gradient kernels over tools/wbench.py's shapes
    if os.path.isdir(dd):
        rec = {}
        for sub in ('linear_wgrad_lds_kernel', 'linear_wgrad_b16_kernel'):
            sm = summarise(dd, sub)
            if sm:
                rec[sub] = {k: sm[k] for k in ('hbm_read_MB_corrected_x2', 'hbm_write_MB', 'traffic_bytes_per_launch') if k in sm}
        json.dump({'_doc': 'averages over every launch of the symbol in tools/wbench.py --bf16 (R = 393 216 rows; the LDS-staged kernel takes the '
                           '128x256, 256x128 and 256x256 maps: 302 / 302 / 403 MB of operands each); FETCH_SIZE x 2 (gfx950) + WRITE_SIZE', 'kernels': rec},
                  open(os.path.join(DST, 'pmc_wgrad.json'), 'w'), indent=1)
    if os.path.isdir(os.path.join(SRC, 'syn')):
        for f in _g.glob(os.path.join(SRC, 'syn', '*')):
            base = os.path.basename(f)
            if base.endswith('kernel_stats.csv'):          # top 40 rows only
                rows = list(csv.DictReader(open(f)))
                with open(os.path.join(DST, 'syn_' + base.replace('_kernel_stats.csv', '_kernel_stats_top40.csv').replace('syn_', '')), 'w', newline='') as g:
                    w = csv.DictWriter(g, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
                    w.writeheader()
                    for r in rows[:40]:
                        w.writerow(r)
            elif base.startswith('bench_'):
                shutil.copy(f, os.path.join(DST, 'rocprof_' + base))
    names += [os.path.basename(f) for f in _g.glob(os.path.join(SRC, 'kbench_*.log'))]
    # round 6 (second half): the synthetic-PGM backward's piece forms, and the no-profiler measurements of the replayed graph
    for sub, prefix in (('ext', 'ext_'), ('graph', 'graph_')):
        for f in _g.glob(os.path.join(SRC, sub, '*.txt')) + _g.glob(os.path.join(SRC, sub, '*.json')):
            if os.path.getsize(f) < 2 * 1024 * 1024:
                shutil.copy(f, os.path.join(DST, prefix + os.path.basename(f)))
    if os.path.exists(os.path.join(SRC, 'ext', 'pmc', 'pmc_ext_bwd.json')):
        shutil.copy(os.path.join(SRC, 'ext', 'pmc', 'pmc_ext_bwd.json'), os.path.join(DST, 'pmc_ext_bwd.json'))
    for f in names:
        if os.path.exists(os.path.join(SRC, f)):
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
    if os.path.exists(os.path.join(SRC, 'timeline', 'step_sequence.csv')):
        shutil.copy(os.path.join(SRC, 'timeline', 'step_sequence.csv'), os.path.join(DST, 'train_step_sequence.csv'))
    for f in ('timeline.txt', 'timeline.json', 'critical_path.txt'):
        if os.path.exists(os.path.join(SRC, 'timeline', f)):
            shutil.copy(os.path.join(SRC, 'timeline', f), os.path.join(DST, 'train_step_' + f))
    if os.path.exists(os.path.join(SRC, 'traffic', 'step_traffic.txt')):
        shutil.copy(os.path.join(SRC, 'traffic', 'step_traffic.txt'), os.path.join(DST, 'train_step_traffic.txt'))
    print(json.dumps(traffic['kernels'], indent=1))


if __name__ == '__main__':
    main()
