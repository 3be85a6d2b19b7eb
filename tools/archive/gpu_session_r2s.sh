#!/bin/bash
# round-2 session s: GPU tests, bench, rocprof passes with the final tree
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2s
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('Ant', d['value']/1e6, d['ms_per_step'], 'pooled', d['pooled']['ms_per_step'])
for k in ('extra','extra2','extra3'):
    print(d[k]['workload'][:40], d[k]['value']/1e6, d[k]['ms_per_step'])
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_r2.sh r2s > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
