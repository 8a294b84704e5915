#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS of every kernel in the device code of ako_plan.hip (cross-compiles, no GPU)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/ako_plan.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only"] + os.environ.get("AKO_HIPCC_EXTRA", "").split() + ["-o", out,
                os.path.join(ROOT, "ako_amd/csrc/ako_plan.hip")], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
filt = sys.argv[1] if len(sys.argv) > 1 else ""
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    if filt not in name:
        continue
    g = lambda k: re.search(r'\.amdhsa_%s (\d+)' % k, body).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:90]:90s} vgpr {g('next_free_vgpr'):>4} sgpr {g('next_free_sgpr'):>4} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
