#!/bin/bash
# Runs on the GPU box (via gpurun): every measurement pass of ONE workload of bench.py, keyed <scene>_<W>x<H>_<mode>.
#   bash tools/gpu_profile_workload.sh dungeon_1920x1080_image --scene dungeon
# Leaves under gpurun_out/prof_<key>/: bench.json (the bench line, with cpu_baseline), stats/ (rocprofv3 --kernel-trace --stats of the
# two-stream run), stats_serial/ (ST_NO_OVERLAP=1: one kernel at a time), fetch/ write/ (--pmc FETCH_SIZE / WRITE_SIZE, SEPARATE passes as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; counters never share a run with --stats), sq/ (8 SQ counters), lane/ (lane
# utilisation: SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)). tools/summarize_profiles.py <tag> --key <key> turns them into the
# tracked files under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
KEY="$1"; shift
OUT=gpurun_out/prof_${KEY}
rm -rf "$OUT"; mkdir -p "$OUT"
python -c 'import __graft_entry__ as g; g.build()' || exit 1
BENCH_EXTRA=${BENCH_EXTRA:-}
timeout 900 python bench.py --no-extras $BENCH_EXTRA "$@" > $OUT/bench.json 2> $OUT/bench.err
Q="--no-cpu-baseline --no-extras --no-profile"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py $Q "$@" > $OUT/stats.log 2>&1
ST_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python bench.py $Q "$@" > $OUT/stats_serial.log 2>&1
# counters: the summary keeps the second half of the launches (24 warm-up + 24 timed frames, all steady state after the pre-roll)
P="--steps 24 --warmup 24 $Q"
ST_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench.py $P "$@" > $OUT/fetch.log 2>&1
ST_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench.py $P "$@" > $OUT/write.log 2>&1
P="--steps 12 --warmup 12 $Q"
ST_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python bench.py $P "$@" > $OUT/sq.log 2>&1
ST_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/lane -- python bench.py $P "$@" > $OUT/lane.log 2>&1
# keep what the summary needs, drop the bulky traces (gpurun_out merges back at most 64 MiB)
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
head -c 300 $OUT/bench.json; echo; du -sh $OUT
