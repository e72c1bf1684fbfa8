#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line of the current build plus the rocprofv3 passes of gpu_profile_round.sh,
# without the test suite, and the FETCH_SIZE / WRITE_SIZE calibration kernels (tools/fetch_calib.hip).
# Extra arguments are passed to every bench.py invocation (e.g. --scene dungeon).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
python bench.py "$@" > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_serial gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/calib_fetch gpurun_out/calib_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/prof_stats.log 2>&1
ST_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_serial -- python bench.py --no-cpu-baseline --no-extras --no-profile "$@" > gpurun_out/prof_stats_serial.log 2>&1
# counters: steady-state frames only would be ideal; the summary keeps the second half of the launches (24 warm-up + 24 timed)
ST_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 24 --warmup 24 --no-cpu-baseline --no-extras --no-profile "$@" > gpurun_out/prof_fetch.log 2>&1
ST_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 24 --warmup 24 --no-cpu-baseline --no-extras --no-profile "$@" > gpurun_out/prof_write.log 2>&1
ST_NO_OVERLAP=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/prof_sq -- python bench.py --steps 12 --warmup 12 --no-cpu-baseline --no-extras --no-profile "$@" > gpurun_out/prof_sq.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null
/tmp/fetch_calib > gpurun_out/calib_truth.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_fetch -- /tmp/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_write -- /tmp/fetch_calib > /dev/null 2>&1
head -c 400 gpurun_out/bench_default.json; echo
