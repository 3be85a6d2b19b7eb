#!/usr/bin/env python3
"""The device probe of bench.py's "box" object by itself, next to a short Ant / ShadowHand timing: which kind of box is this?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print("box:", bench.box_probe("cuda:0"), flush=True)
os.system(f"{sys.executable} {os.path.dirname(os.path.abspath(__file__))}/step_time.py Ant:4096:1500 ShadowHand:16384:300 2>&1 | grep rep2")
os.system("rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -i 'sclk\\|fclk\\|mclk\\|power' | head -8")
