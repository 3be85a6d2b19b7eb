#!/bin/bash
# Instruction-cache counters of the step kernels (separate --pmc pass, no tracing): are the limb / finger waves -- 15 k to 70 k straight-line
# instructions per kernel, four different role bodies per workgroup -- fed from the 64 KB instruction caches or from L2?
# Usage (on the GPU box): tools/profile_icache.sh <tag>  ->  gpurun_out/<tag>_icache.txt
TAG=${1:-ic}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -iE "ICACHE|IFETCH|WAIT_INST|INST_LEVEL|INSTS_ALL|SQ_BUSY_CY" | head -30 > $R/gpurun_out/${TAG}_counters_available.txt
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $OUT -o $TAG -- python $R/tools/step_time.py Ant:4096:300 Humanoid:8192:150 ShadowHand:16384:100 AnymalTerrain:4096:100 > $OUT/log.txt 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in f:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "mi::" in k and ("substep" in k or "post" in k or "pre_kernel" in k or "tips" in k):
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    req, hit, miss = m.get("SQC_ICACHE_REQ", 0), m.get("SQC_ICACHE_HITS", 0), m.get("SQC_ICACHE_MISSES", 0)
    out.append(f"{k:90s} icache req {req:12.0f} hits {hit:12.0f} misses {miss:11.0f} hit-rate {100 * hit / max(req, 1):5.1f} %  wave-cycles {m.get('SQ_WAVE_CYCLES', 0):12.0f} wait-inst {m.get('SQ_WAIT_INST_ANY', 0):12.0f} ifetch {m.get('SQ_IFETCH', 0):10.0f}")
open("$R/gpurun_out/${TAG}_icache.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -3 $OUT/log.txt | cut -c1-200
rm -rf $OUT
