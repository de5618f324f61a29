#!/bin/bash
# Per-kernel VGPR / scratch / occupancy table from the compiler's own remarks (no GPU needed):
#   bash tools/kernel_resources.sh [name-filter]
#   ANET_RES_SOURCE=piece_grad_unit.hip ANET_RES_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp" bash tools/kernel_resources.sh   (the unit built for ILP)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I "$ROOT/include" -c "$ROOT/allocnet_amd/csrc/${ANET_RES_SOURCE:-allocnet_amd.hip}" ${ANET_RES_FLAGS:-} \
  -o /tmp/anet_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys, subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass', line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
names = list(rows)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
for n, d in zip(names, dem):
    if flt and flt not in d: continue
    r = rows[n]
    print('%-72s vgpr %3s agpr %3s scratch %5s occ %s' % (d[:72], r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'), r.get('Occupancy')))
" "$1"
