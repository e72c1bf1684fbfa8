#!/bin/bash
# On the GPU box, after tools/abl_build.sh: the fast build's whole-frame reports (config 3 as written, dungeon 1080p, dungeon 3840x2160) against the
# oracle for the shipped library and for each ablation build, then the frame times of the same libraries (two interleaved rounds).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/abl
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for v in shipped sincos hemi mtdiv sincos_mtdiv; do
  if [ $v = shipped ]; then unset STROLLE_HIP_LIB; else export STROLLE_HIP_LIB=$GRAFT_REPO_ROOT/ab_base/abl_$v.so; fi
  ST_TOL_REPORT_ONLY=1 timeout 900 python -m pytest tests/test_gpu_fast_steady_state.py -q -m gpu -k "config3_as_written or single_step_dungeon_1080p or single_step_dungeon_4k" 2>&1 | tail -2
  for f in gpurun_out/fast_steady_dungeon134k_gi_diffuse_1920x1080.json gpurun_out/fast_steady_dungeon_1920x1080.json gpurun_out/fast_steady_dungeon_3840x2160.json; do
    cp $f gpurun_out/abl/$(basename ${f%.json})__$v.json
  done
done
unset STROLLE_HIP_LIB
python - <<'PY' | tee gpurun_out/r6_abl_gates.txt
import glob, json, os
for f in sorted(glob.glob("gpurun_out/abl/*.json")):
    d = json.load(open(f)); rows = d["whole_frame_rows"]
    filt = [r for r in rows if "psnr" in r]; disc = [r for r in rows if "psnr" not in r]
    wf = max(filt, key=lambda r: r["bad_fraction"]); wp = min(filt, key=lambda r: r["psnr"]); wd = max(disc, key=lambda r: r["bad_fraction"])
    print("%-70s filtered %.2e (%s f%d) psnr %.1f (%s) discrete %.2e (%s)" % (os.path.basename(f)[:-5], wf["bad_fraction"], wf["plane"], wf["frame"], wp["psnr"], wp["plane"], wd["bad_fraction"], wd["plane"]))
PY
for round in 1 2; do for w in cornell dungeon dungeon134k:gi_diffuse; do for v in shipped sincos hemi mtdiv sincos_mtdiv; do
  IFS=: read scene mode <<< "$w"
  if [ $v = shipped ]; then unset STROLLE_HIP_LIB; else export STROLLE_HIP_LIB=$GRAFT_REPO_ROOT/ab_base/abl_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-profile --scene $scene --mode ${mode:-image} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v $w round $round: %.4f ms' % d['ms_per_step'])"
done; done; done 2>&1 | tee gpurun_out/r6_abl_times.txt
