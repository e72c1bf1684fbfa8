cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ph in static animated; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_anim_$ph -- python tools/animated_cost.py --subdivide 2 --refit 2 --all --only $ph > gpurun_out/prof_anim_$ph.log 2>&1
grep "ms/frame" gpurun_out/prof_anim_$ph.log
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_anim_$ph/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("   %-60s %5s x %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_anim_$ph -name "*_kernel_trace.csv" -delete
done
