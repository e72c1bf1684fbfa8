#!/bin/bash
# On the GPU box: the last a-trous pass + composition at 6 waves per SIMD (ST_EXP=0x1000: composition's texels requested with the indirect signal's gathers)
# against the shipped form: frame + the launch's own time, three interleaved rounds; then the fast build's whole-frame gates with the bit set.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
bash tools/gpu_exp.sh 0x1000 "denoise_wavelet+comp" cornell dungeon dungeon:image:3840:2160 2>&1 | tee gpurun_out/r6_far6.txt
ST_EXP=0x1000 timeout 900 python -m pytest tests/test_gpu_fast_steady_state.py -q -m gpu -x -k "whole_frame_single_step and not device_built and not config3" 2>&1 | grep -E "passed|failed" | tee gpurun_out/r6_far6_gates.txt
