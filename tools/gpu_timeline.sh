#!/bin/bash
# Runs on the GPU box (via tools/gpurun_batch.sh): rocprofv3 --kernel-trace of a short two-stream bench run, reduced on the box by
# tools/timeline.py to the timeline of the last frames (which kernels run beside which, how long the chip has 0 / 1 / 2 kernels in flight).
#   bash tools/gpu_timeline.sh <name> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
NAME="$1"; shift
OUT=gpurun_out/timeline_${NAME}
rm -rf "$OUT"; mkdir -p "$OUT"
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline --no-extras --no-profile --steps 12 --warmup 6 "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log | head -c 300; echo
python tools/timeline.py $(find $OUT/trace -name "*_kernel_trace.csv" | head -1) --frames 3 > $OUT/timeline.txt
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
cat $OUT/timeline.txt
