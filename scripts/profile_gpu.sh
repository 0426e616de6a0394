#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of the default bench command (frames of config 2 and config 3;
# the bench's own rocprofv3 PMC children are switched off inside the profiled run: --no-pmc) -> gpurun_out/prof_<tag>/summary.txt.
# The PMC figures (HBM bytes, VALU busy, lane utilisation) come from bench.py itself: its JSON line carries them.
# usage: scripts/profile_gpu.sh <tag> [bench args]
set -u
TAG=${1:-r2}; shift
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-pmc --no-cpu-baseline --no-one-shot "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 - $OUT <<'PY' > $OUT/summary.txt
import glob, json, os, sys
out = sys.argv[1]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-pmc --no-cpu-baseline --no-one-shot " + " ".join(sys.argv[2:]))
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    print("\n## kernel_stats.csv (all dispatches of the run: warm-up, tuning, counting variants <true ...> and timed frames)")
    print(open(f).read().strip())
for line in open(os.path.join(out, "trace.log")):
    if line.startswith("{"):
        j = json.loads(line)
        print("\n## bench line of the profiled run (profiled clocks; not a headline number)")
        keep = {k: j[k] for k in ("value", "unit", "ms_per_step", "steps") if k in j}
        keep["roofline"] = {k: j["roofline"][k] for k in ("achieved", "frac", "kernel", "kernel_ms_avg", "algorithmic_bytes_per_launch", "path")}
        if "config3" in j:
            keep["config3"] = {"ms_per_step": j["config3"]["ms_per_step"], "kernel_ms_avg": j["config3"]["roofline"]["kernel_ms_avg"], "path": j["config3"]["roofline"]["path"]}
        print(json.dumps(keep))
PY
cat $OUT/summary.txt
