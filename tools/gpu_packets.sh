#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_fast_tolerance.py -x -q -m gpu 2>&1 | tail -4
bash tools/gpu_ab_w.sh dungeon dungeon134k:gi_diffuse dungeon:image:3840:2160 2>&1 | sed -E 's/\| .*(prim_visibility [0-9.]+).*/| \1/' | cut -c1-120
