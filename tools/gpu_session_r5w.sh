#!/bin/bash
# round 5: rocprofv3 of the scene kernel under the unmodified FrankaCubeStack task at 4096 envs: kernel-trace stats, then (separately) SQ counters
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/scene_time.py 4096"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r5w -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o r5w -- $CMD > $OUT/pmc_sq.log 2>&1; echo "pmc rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200 > $OUT/kernel_stats_head.csv; cat $OUT/kernel_stats_head.csv
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "scene" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, d in acc.items():
    n = cnt[k]; w = d["SQ_WAVES"] / n
    line = f"{k[:60]}: launches {n}, waves/launch {w:.0f}, per wave: VALU {d['SQ_INSTS_VALU']/d['SQ_WAVES']:.0f}, SALU {d['SQ_INSTS_SALU']/d['SQ_WAVES']:.0f}, LDS {d['SQ_INSTS_LDS']/d['SQ_WAVES']:.0f}, wave cycles (quad) {d['SQ_WAVE_CYCLES']/d['SQ_WAVES']:.0f}, active {d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVES']:.0f}, wait {d['SQ_WAIT_ANY']/d['SQ_WAVES']:.0f}"
    print(line); open("$OUT/scene_pmc.txt", "a").write(line + "\n")
PY
