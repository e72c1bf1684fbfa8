#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_fast_tolerance.py -x -q -m gpu -k "built_on_the_device" 2>&1 | tail -25
