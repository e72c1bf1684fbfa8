#!/bin/bash
# (Round 6, archived experiment: the ST_EXP bits below exist only with tools/experiments/gi_sampling_pool.inc applied to the tree.)
# On the GPU box: the lane-refill pool for the GI bounce rays (ST_EXP 0x100 fused / 0x200 split; bits 12-15: refill threshold / 4) — same bits as the
# plain launch? what do the launches cost? lane utilisation by the SQ counters. -> gpurun_out/r6_pool_*.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('$1: %.4f ms | ' % d['ms_per_step'] + ' '.join('%s %.1f' % (n[:22], k[n].get('us_per_launch_kernel_events', k[n]['us_per_launch'])) for n in k if n.startswith('gi_sampling') or n.startswith('gi_spatial') or n.startswith('prim')))"; }
for round in 1 2; do for w in dungeon dungeon134k:gi_diffuse; do
  IFS=: read scene mode <<< "$w"
  for v in "base::" "pool16:0x100:" "pool32:0x8100:" "base_nv::1" "split16:0x200:1"; do
    IFS=: read name exp nv <<< "$v"
    unset ST_EXP ST_NO_FUSE_GI_VALIDATION
    [ -n "$exp" ] && export ST_EXP=$exp
    [ -n "$nv" ] && export ST_NO_FUSE_GI_VALIDATION=1
    timeout 300 python bench.py --no-cpu-baseline --no-extras --scene $scene --mode ${mode:-image} 2>/dev/null | tail -1 | line "$name $w round $round"
  done
done; done 2>&1 | tee gpurun_out/r6_pool_ab.txt
unset ST_EXP ST_NO_FUSE_GI_VALIDATION
bash tools/gpu_counters.sh dungeon 2>&1 | grep -v amdgpu.ids | sed 's/^/plain: /' | tee gpurun_out/r6_pool_counters.txt
ST_EXP=0x100 bash tools/gpu_counters.sh dungeon 2>&1 | grep -v amdgpu.ids | sed 's/^/pool16: /' | tee -a gpurun_out/r6_pool_counters.txt
ST_EXP=0x200 ST_NO_FUSE_GI_VALIDATION=1 bash tools/gpu_counters.sh dungeon 2>&1 | grep -v amdgpu.ids | sed 's/^/split16: /' | tee -a gpurun_out/r6_pool_counters.txt
