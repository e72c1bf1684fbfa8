#!/bin/bash
# Second pass of the round's evidence (after tools/summarize_profiles.py has written profiles/pmc/<key>.json from the first): the bench lines
# alone, so that each quotes the counter file of the SAME commit (frame_bytes.counter_source names the stamp).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r05}
python -c 'import __graft_entry__ as g; g.build()' || exit 1
run() { key=$1; shift; mkdir -p gpurun_out/prof_$key; timeout 600 python bench.py --no-extras "$@" > gpurun_out/prof_$key/bench.json 2> gpurun_out/prof_$key/bench.err; head -c 200 gpurun_out/prof_$key/bench.json | tail -c 90; echo; }
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; head -c 260 gpurun_out/${TAG}_bench.json | tail -c 90; echo
run cornell_1920x1080_image
run dungeon_1920x1080_image --scene dungeon
run dungeon_3840x2160_image --scene dungeon --width 3840 --height 2160
run dungeon134k_1920x1080_gi_diffuse --scene dungeon134k --mode gi_diffuse
