"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel.
usage: python tools/kusage.py <file.hip> [name-filter]   (compiles device-only into /tmp)"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
       "-fno-slp-vectorize", "-Wno-pass-failed", "-I" + os.path.join(root, "include"), "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kusage.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", dem).replace("void ", "").replace("ofdm::", "")
    if flt and flt not in short:
        continue
    print(f"{short:55s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPRspill {r.get('SGPRs Spill','?'):>3s} "
          f"VGPRspill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} "
          f"LDS {r.get('LDS Size [bytes/block]','?')}")
