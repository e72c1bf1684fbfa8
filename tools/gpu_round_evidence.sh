#!/bin/bash
# Runs on the GPU box (via tools/gpurun_batch.sh): the round's tracked evidence — the driver's bench line, and for each BASELINE.json
# workload that fits one GPU the bench line + every rocprofv3 pass (tools/gpu_profile_workload.sh). Summarised afterwards, locally, by
#   for k in cornell_1920x1080_image dungeon_1920x1080_image dungeon_3840x2160_image dungeon134k_1920x1080_gi_diffuse; do python tools/summarize_profiles.py r04 --key $k; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r05}
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; head -c 400 gpurun_out/${TAG}_bench.json; echo
bash tools/gpu_profile_workload.sh cornell_1920x1080_image
bash tools/gpu_profile_workload.sh dungeon_1920x1080_image --scene dungeon
bash tools/gpu_profile_workload.sh dungeon_3840x2160_image --scene dungeon --width 3840 --height 2160
bash tools/gpu_profile_workload.sh dungeon134k_1920x1080_gi_diffuse --scene dungeon134k --mode gi_diffuse
timeout 600 python bench.py --mode reference --width 3840 --height 2160 --no-profile > gpurun_out/${TAG}_bench_config4_n1.json 2> gpurun_out/${TAG}_bench_config4_n1.err; head -c 300 gpurun_out/${TAG}_bench_config4_n1.json; echo
