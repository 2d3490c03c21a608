"""profiles/rNN_kernel_resource_usage.txt: hipcc's kernel-resource-usage remarks, one row per kernel (needs no GPU)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "alignasm_amd/csrc/aasm_gpu.hip", "-o", "/tmp/aasm_ru.o", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|SGPRs Spill|VGPRs Spill|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name": cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
print("# hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c alignasm_amd/csrc/aasm_gpu.hip -Rpass-analysis=kernel-resource-usage (ROCm 7.2), one row per kernel (tools/resource_usage.py)")
print("%-52s %5s %5s %8s %4s %6s %6s %7s" % ("kernel", "SGPR", "VGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
for r in rows:
    print("%-52s %5s %5s %8s %4s %6s %6s %7s" % (r["name"], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
