#!/usr/bin/env python3
"""Registers / spills / scratch / occupancy / LDS of every kernel of libcasim (clang -Rpass-analysis=kernel-resource-usage on the two
device translation units, same flags as csrc/Makefile).  usage: python tools/kernel_resources.py > profiles/<tag>_kernel_resource_usage.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kubernetes_autoscaler_amd", "csrc")
BASE = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-c", "-o", "/dev/null",
        "-Rpass-analysis=kernel-resource-usage"]
rows = []
for tu, extra in (("casim_engine.hip", []), ("casim_pack_tu.hip", ["-mllvm", "-structurizecfg-skip-uniform-regions=1"])):
    p = subprocess.run(BASE + extra + [tu], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    cur = None
    for line in p.stdout.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*$", "", name)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
print(f"{'kernel':110s} {'sgpr':>5s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s} {'occ':>4s} {'sspill':>6s} {'vspill':>6s} {'lds':>6s}")
for r in rows:
    if "sgpr" not in r:
        continue
    print(f"{r['name'][:110]:110s} {r.get('sgpr', 0):5d} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('scratch', 0):7d} {r.get('occ', 0):4d} "
          f"{r.get('sspill', 0):6d} {r.get('vspill', 0):6d} {r.get('lds', 0):6d}")
