#!/usr/bin/env python3
"""Run tools/explore.py once per library variant found under cann-ops-adv-vllm_amd/lib_var/."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var_dir = os.path.join(ROOT, "cann-ops-adv-vllm_amd", "lib_var")
names = sys.argv[1].split(",") if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else sorted(os.listdir(var_dir))
extra = sys.argv[2:] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else sys.argv[1:]
for n in names:
    lib = os.path.join(var_dir, n, "libcust_opapi.so")
    if not os.path.exists(lib):
        continue
    env = dict(os.environ, SPFA_LIB=lib, SPFA_DEBUG="1", EXPLORE_BNSD="0")
    print(f"=== {n}", flush=True)
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "explore.py")] + extra, env=env,
                       capture_output=True, text=True, timeout=280)
    out = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l]
    print("\n".join(out[-8:]), flush=True)
