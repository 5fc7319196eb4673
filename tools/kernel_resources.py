#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel of the library, from the compiler's own resource-usage remarks (no GPU
needed).  usage: tools/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-shared", "-I", "include", "-I", "rtiow_amd/csrc",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], "-o", "/tmp/_kr.so", "rtiow_amd/csrc/rt_api.hip"]
out = subprocess.run(cmd, cwd=root, capture_output=True, text=True)
cur, rows = None, {}
for line in (out.stderr + out.stdout).splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*?): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
if out.returncode:
    print(out.stderr[-3000:]); sys.exit(out.returncode)
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    g = lambda k: r.get(k, "?")
    print(f"{dem:52s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>4} spillS {g('SGPRs Spill'):>3} spillV {g('VGPRs Spill'):>3} "
          f"scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')} LDS {g('LDS Size [bytes/block]')}")
