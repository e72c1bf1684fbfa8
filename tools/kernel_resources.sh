#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel of the fast (default) or exact build, from the compiler's own remarks
# (-Rpass-analysis=kernel-resource-usage). Runs anywhere hipcc does (no GPU).   tools/kernel_resources.sh [fast|exact] [k_gi k_trace ...]
cd "$(dirname "$0")/../strolle_amd/csrc" || exit 1
B=${1:-fast}; shift
K=${@:-k_trace k_di k_gi k_denoise k_bvh k_util k_atmosphere}
if [ "$B" = fast ]; then F="-ffp-contract=fast-honor-pragmas -DST_KNS=fast -DST_FAST_MATH=1"; else F="-ffp-contract=off -DST_KNS=exact"; fi
for k in $K; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -w $F $EXTRA -Rpass-analysis=kernel-resource-usage -c $k.hip -o /tmp/res_$k.o 2>&1 |
  python3 -c "
import sys, re, subprocess
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([\w][\w \[\]/]*?): (\S+) \[-Rpass', line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
for name, r in rows.items():
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(.*', '', dem).replace('void st::', '')
    print('%-72s vgpr %3s agpr %2s sgpr %3s scratch %4s lds %6s occupancy %s' % (dem[:72], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
"
done
