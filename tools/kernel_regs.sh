#!/bin/bash
# tools/kernel_regs.sh <file.hip> [name pattern] [extra -D flags]: registers / spills / scratch / LDS / occupancy of the gfx950 kernels of
# one source file, from hipcc's own resource-usage remarks (build container, no GPU needed)
F=$1; PAT=${2:-.}; shift; shift
cd $(dirname $0)/../pointreggpt_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 --cuda-device-only -c $F -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
pat = re.compile(sys.argv[1])
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z\[\]/ ]+?): (\S+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, r in rows.items():
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    if not pat.search(dem):
        continue
    print("vgpr %4s agpr %4s spill %3s scratch %5s lds %6s occ %2s  %s" % (r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
          r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?"), dem[:140]))
' "$PAT"
