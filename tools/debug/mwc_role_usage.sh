#!/bin/bash
# Resource usage (registers, spills, scratch) of the Humanoid limb-wave kernels: the whole kernels and one role at a time (-DMI_MWC_ONLY_ROLE).
# Usage: tools/debug/mwc_role_usage.sh [roles...]   (default: all four), extra hipcc flags through MI_EXTRA
CSRC=$(cd $(dirname $0)/../../isaacgymenvs_amd/csrc && pwd)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage ${MI_EXTRA:-}"
ROLES=${@:-0 1 2 3}
( cd $CSRC && /opt/rocm/bin/hipcc $FL -c kernels_humanoid_mwc.hip -o /tmp/khm_all.o > /tmp/khm_all.log 2>&1 ) &
for r in $ROLES; do ( cd $CSRC && /opt/rocm/bin/hipcc $FL -DMI_MWC_ONLY_ROLE=$r -c kernels_humanoid_mwc.hip -o /tmp/khm_r$r.o > /tmp/khm_r$r.log 2>&1 ) & done
wait
python3 - $ROLES <<'PY'
import re, sys
for tag in ["all"] + ["r" + r for r in sys.argv[1:]]:
    f = f"/tmp/khm_{tag}.log"
    txt = open(f).read()
    if "error:" in txt:
        print(tag, txt[-1500:]); continue
    cur, d = None, {}
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m: cur = m.group(1); d[cur] = {}
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
        if m and cur: d[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    for k, v in d.items():
        if "mwc" in k: print(f"{tag:4s} {'fused' if 'fused' in k else 'plain':5s}", v)
PY
