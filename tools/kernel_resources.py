#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel in a .hip file, as the compiler reports them:
     python tools/kernel_resources.py cobbletrace_amd/csrc/ct_pipeline.hip [name filter]"""
import os, re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
here = os.path.dirname(os.path.abspath(src))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-I" + os.path.join(here, "..", "..", "include"),
       "-I" + here, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src] + \
      [a for a in os.environ.get("EXTRA", "").split() if a]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("%-72s %5s %5s %7s %6s %6s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "vspill", "sspill", "occ"))
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("void ", "")
    if flt and flt not in n: continue
    print("%-72s %5d %5d %7d %6d %6d %4d" % (n[:72], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
                                               r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("Occupancy", -1)))
