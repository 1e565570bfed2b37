#!/usr/bin/env python3
"""VGPRs / scratch bytes / occupancy of every kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage,
gfx950), one line per kernel, demangled and sorted by name.  Runs in the CPU container (cross-compile).
    tools/kernel_resources.py file.hip [regex on the demangled name]"""
import os, re, subprocess, sys

src = os.path.abspath(sys.argv[1])
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
extra = ["-fno-slp-vectorize"] if os.path.basename(src) == "kde_kernels.hip" else []   # as csrc/Makefile builds it
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", *extra, "-Rpass-analysis=kernel-resource-usage",
                      "-c", os.path.basename(src), "-o", "/tmp/kr_%d.o" % os.getpid()], cwd=os.path.dirname(src), capture_output=True, text=True).stderr
os.path.exists("/tmp/kr_%d.o" % os.getpid()) and os.remove("/tmp/kr_%d.o" % os.getpid())
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split()[0]] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda rn: rn[1]):
    n = re.sub(r"\(pbn::\w+\)$", "", n.replace("void pbn::", ""))
    if pat.search(n):
        print(f"vgpr {r.get('VGPRs','?'):>4} sgpr {r.get('SGPRs','?'):>4} scratch {r.get('ScratchSize','?'):>4} lds {r.get('LDS','?'):>6} occ {r.get('Occupancy','?')}  {n}")
