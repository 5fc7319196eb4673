#!/usr/bin/env python3
"""Static instruction counts of one kernel by SOURCE LINE (diagnostic, no GPU): compiles the library's device code to
assembly with line info and attributes every instruction to the innermost `.loc`.
usage: tools/isa_lines.py [kernel-substring] [-D...]      default kernel: render_kernelILi5ELb0ELb1E (shipped, small grid)"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = next((a for a in sys.argv[1:] if not a.startswith("-")), "render_kernelILi5ELb0ELb1E")
extra = [a for a in sys.argv[1:] if a.startswith("-")]
asm = "/tmp/_isa_lines.s"
subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", "include", "-I", "rtiow_amd/csrc", "--cuda-device-only", "-S", "-g", *extra,
                "-o", asm, "rtiow_amd/csrc/rt_api.hip"], cwd=root, check=True, stderr=subprocess.DEVNULL)
files, cur, inside = {}, None, False
counts = collections.defaultdict(lambda: collections.Counter())
total = collections.Counter()
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    return "other"
for line in open(asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
        continue
    if re.match(r"^_Z\w+:", line):
        inside = want in line
        continue
    if not inside:
        continue
    if ".end_amdhsa_kernel" in line or line.startswith("\t.size"):
        inside = False
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+([a-z][a-z0-9_]+)", line)
    if m and not line.strip().startswith((".", ";")):
        k = kind(m.group(1))
        counts[cur][k] += 1
        total[k] += 1
print("kernel", want, dict(total))
src = {}
def text(f, l):
    if f not in src:
        try: src[f] = open(os.path.join(root, "rtiow_amd/csrc", f)).read().splitlines()
        except OSError: src[f] = []
    return src[f][l - 1].strip()[:110] if 0 < l <= len(src[f]) else ""
rows = sorted(counts.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"]))
for key, c in rows[:100]:
    f, l = key if key else ("?", 0)
    print(f"{f}:{l:5d} valu {c['valu']:4d} salu {c['salu']:4d} br {c['branch']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} mfma {c['mfma']:2d} | {text(f, l)}")
