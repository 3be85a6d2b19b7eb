#!/usr/bin/env python3
"""Condense a tools/profile_r2.sh output directory (gpurun_out/prof_<tag>) into the tracked profiles/ summaries:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats, verbatim for our kernels + the top torch kernels
   profiles/<tag>_pmc_summary.md     per-kernel averages of the PMC passes (per launch and per wave) + the byte-counter calibration
   profiles/traffic.json             what bench.py reports as roofline.traffic / roofline.valu: per task at its BASELINE size the
                                     calibrated FETCH + WRITE bytes and the executed VALU wave-instructions of ONE control step
Usage: tools/summarize_profile.py <tag>"""
import collections
import csv
import json
import os
import re
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
    for r in rows[1:28]:
        w.writerow([r[0][:160]] + r[1:])
dur_ns = {r[0].split("(")[0].replace("void ", ""): float(r[3]) for r in rows[1:]}


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


out = [f"# rocprofv3 PMC summary `{tag}` (bench.py --steps 400 --warmup 40: Ant@4096, Humanoid@8192, AnymalTerrain@4096, ShadowHand@16384, 1x MI355X)\n",
       "Separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, SQ counters), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.",
       "FETCH_SIZE / WRITE_SIZE are in KB per launch as rocprofv3 reports them; the calibration section turns them into bytes.",
       "SQ counters are summed over all waves of a launch; the per-wave columns divide by SQ_WAVES; SQ_*CYCLES count quad-cycles.\n"]
# ---- calibration on known byte counts in the engine's own access pattern (tools/calib/calib_fetch.hip)
cal = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
K = 32
for name, label in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
    p = os.path.join(src, name, f"{tag}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "soa_" in k:
            per[(k.split("<")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (k, grid), vals in sorted(per.items()):
        n = grid                                   # one thread per env, grid size = N rounded up to 64
        true_kb = K * n * 4 / 1024.0
        steady = vals[1:] if len(vals) > 1 else vals          # first launch after the memset: cold
        cal[label][(k, n)] = (true_kb, sum(steady) / len(steady))
out.append("## Byte-counter calibration (known byte counts, one dword per lane and field, SoA [32][N])\n")
out.append("| kernel | N | true KB read or written per launch | FETCH_SIZE KB | WRITE_SIZE KB |")
out.append("|---|---|---|---|---|")
keys = sorted(set(cal["FETCH_SIZE"]) | set(cal["WRITE_SIZE"]))
for key in keys:
    t = (cal["FETCH_SIZE"].get(key) or cal["WRITE_SIZE"].get(key))[0]
    fv = cal["FETCH_SIZE"].get(key, (0, float("nan")))[1]
    wv = cal["WRITE_SIZE"].get(key, (0, float("nan")))[1]
    out.append(f"| `{key[0]}` | {key[1]} | {t:.0f} | {fv:.1f} | {wv:.1f} |")
big = max([k[1] for k in keys], default=0)
f_fetch = f_write = None
for key in keys:
    if key[1] == big and "soa_copy" in key[0]:
        t = cal["FETCH_SIZE"].get(key, (0, 0))
        if t[1] > 0:
            f_fetch = t[0] / t[1]
        t = cal["WRITE_SIZE"].get(key, (0, 0))
        if t[1] > 0:
            f_write = t[0] / t[1]
out.append(f"\nStreaming factors (largest size, past the 256 MB Infinity Cache): true / FETCH_SIZE = **{f_fetch}**, true / WRITE_SIZE = **{f_write}**.")
out.append("The small size (the engine's own working set: a few hundred KB per field group) stays in L2 between launches, so its counters")
out.append("are far below the byte count: the engine's per-launch FETCH / WRITE values below are what actually crossed the fabric.\n")
f_fetch = f_fetch or 1.0
f_write = f_write or 1.0

traffic = {}
for name, label in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    d, meta = agg(os.path.join(src, name, f"{tag}_counter_collection.csv"))
    for k, v in d.items():
        if ours(k):
            traffic.setdefault(k, {})[label] = sum(v[label]) / len(v[label])
            traffic[k]["meta"] = meta[k]
out.append("| kernel | grid | WG | LDS B/WG | scratch B/lane | VGPR+AGPR | avg us | FETCH_SIZE KB | WRITE_SIZE KB | calibrated KB (F x %.2f + W x %.2f) |" % (f_fetch, f_write))
out.append("|---|---|---|---|---|---|---|---|---|---|")
for k, v in traffic.items():
    m = v["meta"]
    v["bytes"] = 1024.0 * (f_fetch * v.get("FETCH_SIZE", 0) + f_write * v.get("WRITE_SIZE", 0))
    out.append(f"| `{k}` | {m['grid']} | {m['wg']} | {m['lds']} | {m['scratch']} | {m['vgpr']}+{m['agpr']} | {dur_ns.get(k, 0) / 1e3:.1f} | "
               f"{v.get('FETCH_SIZE', 0):.1f} | {v.get('WRITE_SIZE', 0):.1f} | {v['bytes'] / 1024:.1f} |")
d, meta = agg(os.path.join(src, "pmc_sq", f"{tag}_counter_collection.csv"))
out.append("\n| kernel | waves | per-wave: VALU insts | SALU | LDS insts | WAVE_CYCLES (quad) | ACTIVE_INST_ANY | WAIT_ANY | wait % |")
out.append("|---|---|---|---|---|---|---|---|---|")
valu = {}
for k, v in d.items():
    if not ours(k):
        continue
    a = {c: sum(x) / len(x) for c, x in v.items()}
    wv = max(a.get("SQ_WAVES", 1.0), 1.0)
    valu[k] = a.get("SQ_INSTS_VALU", 0.0)
    out.append(f"| `{k}` | {wv:.0f} | {a.get('SQ_INSTS_VALU', 0) / wv:.0f} | {a.get('SQ_INSTS_SALU', 0) / wv:.0f} | {a.get('SQ_INSTS_LDS', 0) / wv:.0f} | "
               f"{a.get('SQ_WAVE_CYCLES', 0) / wv:.0f} | {a.get('SQ_ACTIVE_INST_ANY', 0) / wv:.0f} | {a.get('SQ_WAIT_ANY', 0) / wv:.0f} | "
               f"{100 * a.get('SQ_WAIT_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.0f} |")

# ---- counted fp32 operations and live lanes (round 4; its own pass: tools/profile_r4.sh pmc_flops)
flops, live = {}, {}
pf = os.path.join(src, "pmc_flops", f"{tag}_counter_collection.csv")
if os.path.exists(pf):
    d2, _ = agg(pf)
    out.append("\n| kernel | VALU wave-insts | ADD_F32 | MUL_F32 | FMA_F32 | TRANS_F32 | fp32 share of VALU | live lanes per VALU inst (THREAD_CYCLES / ACTIVE_INST) | FLOP per launch |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    for k, v in d2.items():
        if not ours(k):
            continue
        a = {c: sum(x) / len(x) for c, x in v.items()}
        ll = min(a.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(a.get("SQ_ACTIVE_INST_VALU", 1.0), 1.0), 64.0)
        fins = a.get("SQ_INSTS_VALU_ADD_F32", 0) + a.get("SQ_INSTS_VALU_MUL_F32", 0) + a.get("SQ_INSTS_VALU_TRANS_F32", 0) + a.get("SQ_INSTS_VALU_FMA_F32", 0)
        fl = (a.get("SQ_INSTS_VALU_ADD_F32", 0) + a.get("SQ_INSTS_VALU_MUL_F32", 0) + a.get("SQ_INSTS_VALU_TRANS_F32", 0) + 2 * a.get("SQ_INSTS_VALU_FMA_F32", 0)) * ll
        flops[k], live[k] = fl, ll
        out.append(f"| `{k}` | {a.get('SQ_INSTS_VALU', 0):.0f} | {a.get('SQ_INSTS_VALU_ADD_F32', 0):.0f} | {a.get('SQ_INSTS_VALU_MUL_F32', 0):.0f} | {a.get('SQ_INSTS_VALU_FMA_F32', 0):.0f} | "
                   f"{a.get('SQ_INSTS_VALU_TRANS_F32', 0):.0f} | {100 * fins / max(a.get('SQ_INSTS_VALU', 1), 1):.0f} % | {ll:.1f} | {fl / 1e6:.1f} M |")

# ---- one control step per task = these launches (pattern, launches per step)
RECIPE = {
    # (a fused-sub-step kernel, when the trace has one, is the step's ONE physics launch: option fused_sub, csrc/mw_kernels.hpp)
    # (round 4: with option fused_post the Ant's launch is substep_mw_fused_post_kernel and there is no post kernel in the trace)
    "Ant@4096": [(r"substep_mw_fused_post_kernel<ModelAnt|substep_mw_fused_kernel<ModelAnt|substep(_mw)?_kernel<ModelAnt", {"fused": 1, "plain": 2}), (r"loco_post_kernel<ModelAnt", 1)],
    # (round 4, Humanoid with fused_post: one plain limb-wave launch + one that carries the post step; no post kernel in the trace)
    "Humanoid@8192": [(r"substep_mwc_post_kernel<ModelHumanoid", 1), (r"substep_(sc2_|mwc_)?kernel<ModelHumanoid", {"plain": 2, "with_post": 1}), (r"loco_post_kernel<ModelHumanoid", 1)],
    "AnymalTerrain@4096": [(r"substep_mw_fused_kernel<ModelAnymal, mi::HeightfieldGround|substep(_mw)?_kernel<ModelAnymal, mi::HeightfieldGround", {"fused": 1, "plain": 5}),
                           (r"anymal_post_kernel", 1), (r"anymal_heights_kernel", 1),
                           (r"anymal_cmdnorm_kernel", 1)],
    # (round 4: hand_pre4_kernel -- four lanes per env -- replaces hand_pre_kernel, the post kernel's fingertip groups replace hand_tips_kernel)
    # (round 6: the finger-wave launch comes in two instantiations -- <.., false> for every sub-step of a call but the last, <.., true> with the pairs'
    #  forces on the fingertip sensors for the last: one of each per control step; the one-wave form keeps its single name)
    "ShadowHand@16384": [(r"hand_pre4?_kernel", 1), (r"hand_substep_kernel<(mi::ShadowHandTask, )?0>", 2),
                         (r"hand_substep(_mw64|_mw)_kernel<(mi::ShadowHandTask, )?0, false>", 1), (r"hand_substep(_mw64|_mw)_kernel<(mi::ShadowHandTask, )?0, true>", 1),
                         (r"hand_tips_kernel", 1), (r"hand_post_kernel", 1),
                         (r"hand_finalize_kernel", 1)],
}
tj = {}
out.append("\n## One control step (what bench.py reports as roofline.traffic / roofline.valu)\n")
out.append("| task | launches | calibrated HBM-side bytes / step | VALU wave-instructions / step | kernel time / step (us) |")
out.append("|---|---|---|---|---|")
for task, recipe in RECIPE.items():
    tot_b, tot_v, tot_t, tot_f, names, dom = 0.0, 0.0, 0.0, 0.0, [], (0.0, None)
    for pat, cnt in recipe:
        keys = sorted(traffic, key=lambda k: "fused" not in k)       # a fused kernel first
        for k in keys:
            if re.search(pat, k):
                if isinstance(cnt, dict):
                    if "with_post" in cnt:
                        cnt = cnt["with_post" if any("substep_mwc_post_kernel" in x for x in traffic) else "plain"]
                    else:
                        cnt = cnt["fused" if "fused" in k else "plain"]
                tot_b += cnt * traffic[k]["bytes"]; tot_v += cnt * valu.get(k, 0.0); tot_t += cnt * dur_ns.get(k, 0.0) / 1e3
                tot_f += cnt * flops.get(k, 0.0)
                if cnt * dur_ns.get(k, 0.0) > dom[0]:
                    dom = (cnt * dur_ns.get(k, 0.0), k)
                names.append(f"{cnt} x {k.split('<')[0].replace('mi::', '')}")
                break
    if tot_b > 0:
        tj[task] = {"traffic_bytes_per_step": int(tot_b), "valu_wave_insts_per_step": int(tot_v), "kernel_us_per_step": round(tot_t, 1),
                    "fp32_flops_per_step": int(tot_f) if tot_f > 0 else None, "live_lanes": round(live[dom[1]], 1) if dom[1] in live else None,
                    "source": f"profiles/{tag}_pmc_summary.md (FETCH_SIZE x {f_fetch:.2f} + WRITE_SIZE x {f_write:.2f}, calibrated on tools/calib/calib_fetch.hip)"}
        out.append(f"| {task} | {', '.join(names)} | {tot_b / 1e6:.2f} MB | {tot_v / 1e6:.2f} M | {tot_t:.1f} |")
try:       # the library the passes ran on (tools/profile_r5.sh): bench.py withholds these numbers from a run that loads another build
    tj["_lib_sha256"] = open(os.path.join(src, "lib_sha256.txt")).read().strip()
except OSError:
    pass
json.dump(tj, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_pmc_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
