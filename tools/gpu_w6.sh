#!/bin/bash
# On the GPU box: k_gi_preview_both at 6 waves per SIMD (ST_EXP=0x800: Hit decoded after the two passes, history fetched after them) against the shipped form —
# frame + the launch's own time, three interleaved rounds, then the bit-exact suite's Image-mode tests with the bit set.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
bash tools/gpu_exp.sh 0x800 "gi_preview x2" cornell dungeon dungeon:image:3840:2160 dungeon134k:gi_diffuse 2>&1 | tee gpurun_out/r6_w6.txt
ST_EXP=0x800 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "image or config_2 or scheduling or moving" 2>&1 | grep -E "passed|failed" | tee gpurun_out/r6_w6_parity.txt
