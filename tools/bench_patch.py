"""A/B of module-level scheduling choices without an environment switch:
    python tools/bench_patch.py "<python statements run after importing fgnn_amd: A = mpnn.assemblies, ops, blocks, pointwise>" [bench.py args...]
e.g.  python tools/bench_patch.py "A._V2V_MAIN = {4}" --no-cpu-baseline --steps 30      (round 6: the stream placements of profiles/r06/README.md)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
code = sys.argv[1]
sys.argv = ['bench.py'] + sys.argv[2:]
import bench
import fgnn_amd
from fgnn_amd import ops
from fgnn_amd.mpnn import assemblies as A, blocks, pointwise
exec(code)
bench.main()
