# Shader clock each big kernel actually runs at: GRBM_GUI_ACTIVE (graphics-clock cycles the GPU was busy in the launch) over the launch's
# duration from the kernel trace of the same pass.  One config-3 proof + one interpolate: gpurun -- bash tools/pmc_clock.sh
set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_clock.sh}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_clock; mkdir -p $O
export PIL2GL_BENCH_FROM_HOST=0
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1
python3 - <<PY
import csv, collections, glob
dur = {}
for f in glob.glob("$O/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(list)
for f in glob.glob("$O/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        d = dur.get(r["Dispatch_Id"])
        if d and d[0] > 2_000_000:                       # launches of 2 ms and more
            acc[d[1][:90]].append((float(r["Counter_Value"]), d[0]))
print("kernel, launches >= 2 ms, GRBM_GUI_ACTIVE / duration (GHz; the counter is summed over the 8 XCDs when it comes out ~8x)")
for k, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    g = [c / ns for c, ns in v]
    print("%-92s %3d  median %.3f  min %.3f  max %.3f   (ms: %.1f)" % (k, len(v), sorted(g)[len(g) // 2], min(g), max(g), sorted(x[1] for x in v)[len(v) // 2] / 1e6))
PY
