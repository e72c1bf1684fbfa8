#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, the default bench line, and the rocprofv3 passes whose summaries
# tools/summarize_profiles.py turns into profiles/<tag>_*. Counter passes are separate runs, never combined with tracing
# other than --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_serial gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
# the same command as the bench line (region 1 pipelined on two streams, region 2 serial with event pairs)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_stats.log 2>&1
# serial graph only: per-kernel durations without a second kernel sharing the chip
ST_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_serial -- python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/prof_stats_serial.log 2>&1
ST_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 12 --warmup 12 --no-cpu-baseline --no-extras --no-profile > gpurun_out/prof_fetch.log 2>&1
ST_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 12 --warmup 12 --no-cpu-baseline --no-extras --no-profile > gpurun_out/prof_write.log 2>&1
ST_NO_OVERLAP=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/prof_sq -- python bench.py --steps 12 --warmup 12 --no-cpu-baseline --no-extras --no-profile > gpurun_out/prof_sq.log 2>&1
cat gpurun_out/bench_default.json | head -c 600
echo
# the changing-scene numbers DESIGN.md quotes (host refresh + render; not part of the bench line)
{ python tools/tick_cost.py --device -1 --subdivide 2; python tools/tick_cost.py --device -1 --subdivide 2 --refit; python tools/tick_cost.py --device -1 --subdivide 2 --refit --all;
  python tools/animated_cost.py --subdivide 2 --refit | tail -1; python tools/animated_cost.py --cornell --light | tail -1; } > gpurun_out/animated.log 2>&1; cat gpurun_out/animated.log
