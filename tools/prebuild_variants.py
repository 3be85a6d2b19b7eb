#!/usr/bin/env python3
"""Rebuild the cached run-time asset variants (isaacgymenvs_amd/_variants/<hash>/) whose library is older than the stock one, for HIP and / or CPU,
in this container (hipcc cross-compiles gfx950 without a GPU) -- so that a gpurun session that runs the stand-in / run-time-asset tests on the HIP
backend does not spend its GPU minutes compiling.  Usage: tools/prebuild_variants.py [hip] [cpu] [--all]
--all: also build the flavour a variant directory does not have yet (a variant the CPU tests created here is then ready for the GPU box)."""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgymenvs_amd import native  # noqa: E402
from isaacgymenvs_amd.assets import runtime  # noqa: E402

build_missing = "--all" in sys.argv
kinds = [a for a in sys.argv[1:] if a != "--all"] or ["hip"]
csrc = os.path.join(os.path.dirname(native.__file__), "csrc")
for vdir in sorted(glob.glob(os.path.join(runtime.VARIANT_DIR, "*"))):
    gen = os.path.join(vdir, "pkg", "csrc", "gen")
    model = None
    for h in sorted(glob.glob(os.path.join(gen, "model_*.h"))):
        if open(h).read() != open(os.path.join(csrc, "gen", os.path.basename(h))).read():
            model, txt = os.path.basename(h)[len("model_"):-2], open(h).read()
    if model is None:
        continue
    for kind in kinds:
        cpu = kind == "cpu"
        out = os.path.join(vdir, "libmi_engine_cpu.so" if cpu else "libmi_engine.so")
        stock = native.CPU_LIB_PATH if cpu else native.LIB_PATH
        if not os.path.exists(out) and not build_missing:
            continue                              # this variant was never asked for on that backend
        if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(stock):
            print(f"{os.path.basename(vdir)} {model} {kind}: up to date")
            continue
        t0 = time.time()
        runtime._build_variant(model, txt, vdir, out, cpu, False)
        print(f"{os.path.basename(vdir)} {model} {kind}: rebuilt in {time.time() - t0:.0f} s", flush=True)
