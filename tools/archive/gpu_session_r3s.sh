set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3s/pytest_gpu.log 2>&1; tail -5 gpurun_out/r3s/pytest_gpu.log
timeout 600 python tools/ant_fused_post_ab.py 4096 8192 > gpurun_out/r3s/ant_fused_post_ab.txt 2>&1; cat gpurun_out/r3s/ant_fused_post_ab.txt
timeout 600 python bench.py --steps 1000 --warmup 100 > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r3s/bench.json').read().strip().splitlines()[-1])
print('Ant', d['value']/1e6, d['ms_per_step'], 'pooled', d['pooled']['ms_per_step'], 'kern', d['roofline']['kernel_ms'])
for k in ('extra','extra2','extra3'):
    print(d[k]['workload'][:40], d[k]['value']/1e6, d[k]['ms_per_step'], d[k]['roofline']['kernel_ms'], d[k]['consistent'])
"
