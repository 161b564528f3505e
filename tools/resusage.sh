#!/bin/bash
# per-kernel register / spill / occupancy / LDS summary of one csrc/*.hip file (device-only compile, no GPU needed)
# usage: tools/resusage.sh i2r_encoder [extra hipcc flags]
R=$(cd $(dirname $0)/.. && pwd)
C=$R/intra-and-inter-human-relation-network-for-mpee_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I $R/include -I $C --cuda-device-only -c $C/$f.hip -o /tmp/$f.dev.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c '
import sys, re, subprocess
cur = {}
def flush():
    if cur:
        nm = subprocess.run(["c++filt", cur["Function Name"]], capture_output=True, text=True).stdout.strip()
        nm = re.sub(r"\(anonymous namespace\)::|\(.*\)$|void ", "", nm)
        print("%-48s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %6s" % (nm, cur.get("VGPRs"), cur.get("AGPRs"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
for line in sys.stdin:
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        k = "Function Name"
        flush(); cur = {}
    cur[k] = v
flush()
'
