#!/usr/bin/env python3
"""Resource usage of every kernel of dust_amd/csrc/*.hip as the compiler reports it (hipcc -Rpass-analysis=kernel-resource-usage):
VGPRs, SGPRs, scratch, spills, waves per SIMD. Runs here (cross-compile, no GPU). usage: code_objects.py rNN > profiles/rNN_code_objects.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dust_amd", "csrc")
tag = sys.argv[1] if len(sys.argv) > 1 else "rNN"
print(f"# dust_amd/csrc/*.hip code objects (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage), {tag}")
print(f"{'kernel':<58}{'VGPR':>5}{'SGPR':>6}{'scratch B/lane':>15}{'SGPR spill':>11}{'VGPR spill':>11}{'waves/SIMD':>11}")
for src in ("kernels.hip", "gi.hip", "radix.hip", "edit.hip", "denoise.hip"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", ".", "--cuda-device-only",
                          "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(dust::FrameArgs\)$", "", name).replace("void ", "")
            cur = rows.setdefault(name, {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    for name, r in rows.items():
        if "VGPRs" not in r:
            continue
        print(f"{name[:56]:<58}{r['VGPRs']:>5}{r.get('TotalSGPRs', 0):>6}{r.get('ScratchSize [bytes/lane]', 0):>15}{r.get('SGPRs Spill', 0):>11}"
              f"{r.get('VGPRs Spill', 0):>11}{r.get('Occupancy [waves/SIMD]', 0):>11}")
