#!/usr/bin/env python3
"""Per-kernel instruction statistics of a translation unit's gfx950 ISA: instructions, v_readfirstlane (waterfall loops of dynamic register
indexing), scalar-condition branches and loop headers.  The AllegroHand sub-step once carried 8012 v_readfirstlane and 258 k lines of ISA --
constexpr table look-ups evaluated at run time (profiles/r3z_allegro_hand_fix.txt) -- and nothing but a stop watch noticed; this is the look
that would have.  Usage: tools/isa_scan.py kernels_mw_ant [kernels_anymal ...]   (compiles each with -S; minutes for the big ones)"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "isaacgymenvs_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-signed-zeros", "-fno-trapping-math", "-fno-slp-vectorize", "-S", "--cuda-device-only"]


def scan(path):
    cur, stats = None, {}
    for line in open(path):
        m = re.match(r"^(_ZN2mi[^:]+):", line)
        if m and "kernel" in m.group(1):
            cur = m.group(1)
            stats[cur] = [0, 0, 0, 0]
            continue
        if cur is None:
            continue
        if line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;"):
            stats[cur][0] += 1
            stats[cur][1] += "v_readfirstlane" in line
            stats[cur][2] += "s_cbranch_scc" in line
        stats[cur][3] += "Loop Header" in line
    return stats


for tu in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, tu + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [tu + ".hip", "-o", out], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        for k, v in sorted(scan(out).items(), key=lambda kv: -kv[1][0]):
            if v[0] > 2000:
                print(f"{tu}: {v[0]:7d} instr  readfirstlane {v[1]:5d}  scc-branches {v[2]:4d}  loops {v[3]:4d}  {k[:110]}")
