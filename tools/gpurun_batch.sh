#!/bin/bash
# Local wrapper around gpurun: stamps the snapshot with the commit it is taken from (the GPU box has no .git; bench.py prints the
# stamp, tools/summarize_profiles.py copies it into the counter summaries), then runs the given command on the box.
#   tools/gpurun_batch.sh 1500 'bash tools/gpu_batch.sh tests bench'
cd "$(dirname "$0")/.." || exit 1
mkdir -p profiles
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet HEAD -- strolle_amd include bench.py || echo +dirty) $(date -u +%Y-%m-%dT%H:%MZ)" > profiles/.build_stamp
T=${1:-1200}; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
