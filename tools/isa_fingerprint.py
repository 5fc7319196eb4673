#!/usr/bin/env python3
"""Static fingerprint of the product kernels' ISA (no GPU): per instantiation of render_kernel, the number of instructions by class
(vector / scalar / branch / LDS / vector memory / MFMA) and a hash over the opcode sequence.  A source refactoring that is meant to
leave the machine code alone is checked with it: tools/isa_fingerprint.py > before.txt ... > after.txt; diff."""
import collections, hashlib, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tools"))
import isa_census as ic
elf, dis = ic.build([a for a in sys.argv[1:] if a.startswith("-D")])
cur, seqs = None, collections.OrderedDict()
for line in dis.splitlines():
    m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
    if m:
        cur = m.group(2)
        seqs[cur] = []
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s", line)
    if m and cur:
        seqs[cur].append(m.group(1))
for name, ops in seqs.items():
    if "render_kernel" not in name and "resolve" not in name:
        continue
    c = collections.Counter(ic.classify(o, "")[0].split()[0] for o in ops)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{dem:55s} n={len(ops):5d} valu={c['valu']:5d} salu={c['salu']:5d} branch={c['branch']:4d} lds={c['lds']:4d} vmem={c['vmem']:3d} mfma={c['mfma']:3d} "
          f"ops-sha={hashlib.sha256(' '.join(ops).encode()).hexdigest()[:12]}")
