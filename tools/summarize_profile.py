#!/usr/bin/env python3
"""Condense a tools/profile_r1.sh output directory (gpurun_out/prof_<tag>) into the tracked profiles/ summaries:
   profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, verbatim for our kernels + top torch kernels)
   profiles/<tag>_pmc_summary.md     (per-kernel averages of the PMC passes, per launch and per wave)
Usage: tools/summarize_profile.py <tag>"""
import collections
import csv
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

rows = list(csv.reader(open(os.path.join(src, "trace", f"{tag}_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:26]:
        w.writerow([r[0][:160]] + r[1:])

def ours(k):
    return k.startswith("mi::") or ("_kernel" in k and "at::" not in k and "elementwise" not in k)


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = dict(grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]), lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]),
                       vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]))
    return d, meta

out = [f"# rocprofv3 PMC summary `{tag}` (bench.py --steps 300 --warmup 50, Ant@4096, Humanoid@8192, AnymalTerrain@4096, ShadowHand@16384, 1x MI355X)\n",
       "Separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, SQ counters), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.",
       "FETCH_SIZE / WRITE_SIZE are reported in KB per launch; `fetch_x2` applies the guide's gfx950 correction (the counter tallies",
       "128-B requests at 64 B).  SQ counters are summed over all waves of a launch; the per-wave column divides by SQ_WAVES;",
       "SQ_*CYCLES count quad-cycles (x4 = shader cycles).\n"]
traffic = {}
for name, label in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    d, meta = agg(os.path.join(src, name, f"{tag}_counter_collection.csv"))
    for k, v in d.items():
        if ours(k):
            traffic.setdefault(k, {})[label] = sum(v[label]) / len(v[label])
            traffic[k]["meta"] = meta[k]
out.append("| kernel | grid | LDS B/WG | scratch B/lane | VGPR+AGPR | FETCH_SIZE KB | fetch_x2 KB | WRITE_SIZE KB |")
out.append("|---|---|---|---|---|---|---|---|")
for k, v in traffic.items():
    m = v["meta"]
    out.append(f"| `{k}` | {m['grid']} | {m['lds']} | {m['scratch']} | {m['vgpr']}+{m['agpr']} | {v.get('FETCH_SIZE', 0):.1f} | "
               f"{2 * v.get('FETCH_SIZE', 0):.1f} | {v.get('WRITE_SIZE', 0):.1f} |")
d, meta = agg(os.path.join(src, "pmc_sq", f"{tag}_counter_collection.csv"))
out.append("\n| kernel | waves | per-wave: VALU insts | SALU | LDS insts | WAVE_CYCLES (quad) | ACTIVE_INST_ANY | WAIT_ANY | wait % |")
out.append("|---|---|---|---|---|---|---|---|---|")
for k, v in d.items():
    if not ours(k):
        continue
    a = {c: sum(x) / len(x) for c, x in v.items()}
    wv = max(a.get("SQ_WAVES", 1.0), 1.0)
    out.append(f"| `{k}` | {wv:.0f} | {a.get('SQ_INSTS_VALU', 0) / wv:.0f} | {a.get('SQ_INSTS_SALU', 0) / wv:.0f} | {a.get('SQ_INSTS_LDS', 0) / wv:.0f} | "
               f"{a.get('SQ_WAVE_CYCLES', 0) / wv:.0f} | {a.get('SQ_ACTIVE_INST_ANY', 0) / wv:.0f} | {a.get('SQ_WAIT_ANY', 0) / wv:.0f} | "
               f"{100 * a.get('SQ_WAIT_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.0f} |")
open(os.path.join(dst, f"{tag}_pmc_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
