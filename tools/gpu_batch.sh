#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/gpu_batch.sh <steps...>'): a batch of independent measurement steps, each
# logging to gpurun_out/<tag>_<step>.log. Steps: probe steady tests bench dungeon config5 ab:<flags> profile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r03}
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/${TAG}_build.log 2>&1 || { tail -20 gpurun_out/${TAG}_build.log; exit 1; }
for step in "$@"; do
  echo "=== $step"; t0=$(date +%s)
  case "$step" in
    probe)
      hipcc --offload-arch=gfx950 -O3 tools/tap_probe.hip -o /tmp/tap_probe && timeout 120 /tmp/tap_probe > gpurun_out/${TAG}_tap_probe.txt 2>&1; cat gpurun_out/${TAG}_tap_probe.txt ;;
    steady)
      ST_TOL_REPORT_ONLY=${REPORT_ONLY:-1} timeout 1500 python -m pytest tests/test_gpu_fast_steady_state.py -x -q --durations=8 > gpurun_out/${TAG}_steady.log 2>&1; tail -15 gpurun_out/${TAG}_steady.log ;;
    pytest:*)   # pytest:<name>:<file or dir>:<-k expression>
      spec="${step#pytest:}"; name="${spec%%:*}"; rest="${spec#*:}"; what="${rest%%:*}"; expr="${rest#*:}"
      ST_TOL_REPORT_ONLY=${REPORT_ONLY:-0} timeout 1800 python -m pytest "$what" -m gpu -x -q -k "$expr" --durations=6 > gpurun_out/${TAG}_pytest_${name}.log 2>&1; tail -12 gpurun_out/${TAG}_pytest_${name}.log ;;
    newtests)
      timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "present_copy or bench_two_rank or bench_strong or c_example" --durations=8 > gpurun_out/${TAG}_newtests.log 2>&1; tail -8 gpurun_out/${TAG}_newtests.log ;;
    tests)
      timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -25 gpurun_out/${TAG}_pytest_gpu.log ;;
    bench)
      timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; head -c 1500 gpurun_out/${TAG}_bench.json; echo; tail -3 gpurun_out/${TAG}_bench.err ;;
    dungeon)
      timeout 600 python bench.py --scene dungeon --no-cpu-baseline > gpurun_out/${TAG}_bench_dungeon.json 2> gpurun_out/${TAG}_bench_dungeon.err; head -c 600 gpurun_out/${TAG}_bench_dungeon.json; echo ;;
    config5)
      timeout 600 python bench.py --scene dungeon --width 3840 --height 2160 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_config5_n1.json 2> gpurun_out/${TAG}_bench_config5_n1.err; head -c 400 gpurun_out/${TAG}_bench_config5_n1.json; echo
      timeout 600 python bench.py --mode reference --width 3840 --height 2160 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_config4_n1.json 2> gpurun_out/${TAG}_bench_config4_n1.err; head -c 400 gpurun_out/${TAG}_bench_config4_n1.json; echo ;;
    env:*)   # env:<name>:<VAR=1,VAR2=1>[:bench args]  — the default bench command under environment switches
      spec="${step#env:}"; name="${spec%%:*}"; rest="${spec#*:}"; vars="${rest%%:*}"; extra=""; [ "$rest" != "$vars" ] && extra="${rest#*:}"
      env $(echo "$vars" | tr ',' ' ') timeout 600 python bench.py --no-cpu-baseline --no-extras $extra > gpurun_out/${TAG}_bench_${name}.json 2> gpurun_out/${TAG}_bench_${name}.err
      python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_${name}.json").read().strip().splitlines()[-1])
print("${name}: %.4f ms/frame, %.1f Mray/s, roofline frac %s" % (d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac")))
for k, v in d.get("kernels", {}).items(): print("   %-48s %7.1f us x%.2f" % (k, v["us_per_launch"], v["launches_per_frame"]))
PY
      ;;
    ab:*)
      bash tools/ab_bench.sh "${step#ab:}" ${AB_ARGS---no-extras} > gpurun_out/${TAG}_ab.log 2>&1; cat gpurun_out/${TAG}_ab.log ;;
    refitcost)   # host tick + frame time with moving instances: rebuild / host refit / device refit, 208 k-triangle dungeon
      { for m in 0 1 2; do timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit $m; done
        for m in 1 2; do timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit $m --all; done
        for m in 1 2; do timeout 300 python tools/animated_cost.py --subdivide 2 --refit $m; done
        for m in 1 2; do timeout 300 python tools/animated_cost.py --refit $m; done; } > gpurun_out/${TAG}_refit_cost.txt 2>&1; cat gpurun_out/${TAG}_refit_cost.txt ;;
    profile)
      bash tools/gpu_profile_quick.sh > gpurun_out/${TAG}_profile.log 2>&1; tail -5 gpurun_out/${TAG}_profile.log ;;
    *) echo "unknown step $step" ;;
  esac
  echo "--- $step took $(( $(date +%s) - t0 )) s"
done
