#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_fast_tolerance.py tests/test_gpu_fast_steady_state.py -x -q -m gpu -k "cornell or whole_frame or launches or every_launch or packets or heatmap" 2>&1 | tail -5
bash tools/gpu_ab_w.sh cornell cornell:image:3840:2160 2>&1 | sed -E 's/\| .*(prim_visibility [0-9.]+).*/| \1/' | cut -c1-120
