#!/bin/bash
# Round 6, third call: (1) the tree against the library of the call before (ab_base/base.so): far wavelet with 32-bit tap offsets, the late preview launch's
# speculative tap loads; (2) ST_EXP=0x400 — scenes that fit LDS walk their WIDE stream from LDS with the exact leaf test — against the default on the headline,
# frame + per-kernel, and under the fast build's gates.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
# the fast build's whole-frame gates with the primary hits exact: headroom before / after by tools/gate_headroom.py
timeout 1800 python -m pytest tests/test_gpu_fast_steady_state.py -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6c_steady.txt
timeout 900 python -m pytest tests/test_gpu_fast_tolerance.py tests/test_gpu_parity.py -q -m gpu -x -k "packets or wide_stream or heatmap or reference_mode_psnr or every_launch or config_2 or cornell" 2>&1 | tail -5 | tee -a gpurun_out/r6c_steady.txt
bash tools/gpu_ab_w.sh cornell dungeon dungeon134k:gi_diffuse 2>&1 | tee gpurun_out/r6c_ab.txt
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('$1: %.4f ms moving %s | ' % (d['ms_per_step'], d.get('ms_per_step_moving')) + ' '.join('%s %.1f' % (n[:18], k[n].get('us_per_launch_kernel_events', k[n]['us_per_launch'])) for n in k))"; }
for round in 1 2 3; do for v in off on; do
  if [ $v = on ]; then export ST_EXP=0x400; else unset ST_EXP; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | line "wide-in-LDS $v cornell round $round"
done; done 2>&1 | tee gpurun_out/r6c_wide_lds.txt
export ST_EXP=0x400
timeout 1500 python -m pytest tests/test_gpu_fast_steady_state.py tests/test_gpu_fast_tolerance.py -q -m gpu -k "launches_1080p or whole_frame_single_step or light_and_camera_moving or every_launch_within or reference_mode_psnr or heatmap_integers or image_mode_statistics or report_only" 2>&1 | tail -8 | tee gpurun_out/r6c_wide_lds_gates.txt
unset ST_EXP
mkdir -p gpurun_out/wide_lds && cp gpurun_out/fast_steady_cornell*.json gpurun_out/wide_lds/ 2>/dev/null
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && timeout 60 /tmp/valu_rate | grep -E "u64|pk_|v_fma_f32 " | tee gpurun_out/r6c_valu_rate_extra.txt
