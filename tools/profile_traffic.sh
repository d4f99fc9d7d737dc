#!/bin/sh
# Run on the GPU box: HBM-side bytes of a whole bench.py training step (FETCH_SIZE and WRITE_SIZE, each its own rocprofv3 run).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/traffic
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tf_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tf_$c -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline $FGNN_BENCH_ARGS > /tmp/tf_$c.log 2>&1
  f=$(find /tmp/tf_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY' | tee $R/gpurun_out/traffic/$c.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    tot[r['Kernel_Name']] += float(r['Counter_Value']); cnt[r['Kernel_Name']] += 1
# executed model steps (warm-up, replays and bench.py's two eager instrumented steps alike): every forward opens with the two
# edge-type MLP launches; the optimizer runs outside some of them, so its launch count undercounts
adam = sum(v for k, v in cnt.items() if 'edge_mlp_fwd_kernel' in k) // 2 or cnt.get('flat_adam_kernel(FaParams)', 0)
print(sys.argv[2], 'rows', len(rows), 'executed model steps (edge_mlp_fwd launches / 2)', adam)
allkb = sum(tot.values())
print('total KB over the run: %.0f  -> per executed step (all kernels / adam launches): %.1f MB' % (allkb, allkb / max(adam, 1) / 1024))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print('  %10.1f MB/step  %5.1f launches/step  %s' % (v / max(adam, 1) / 1024, cnt[k] / max(adam, 1), k[:90]))
PY
done
