#!/bin/bash
# Profiles of the bench workload on the GPU box: kernel stats + the two HBM counter passes (separate runs, as
# MI355X_MICROARCH.md section HBM prescribes), condensed into profiles/<tag>_*.
#   gpurun -- 'bash scripts/profile_c3.sh r02_c3 [extra bench.py args]'
# then copy gpurun_out/profiles_<tag>/* into profiles/.
set -e
TAG=${1:-r02_c3}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=3; WARM=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python3 "$REPO/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
# (bench.py --steps 1 --warmup 0 runs TWO steps: the timed one and one for the per-group breakdown)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o p -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o p -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> "$OUT/write.err"
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
T=$(find "$OUT/stats" -name '*kernel_trace.csv' | head -1)
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
cd "$REPO"
python3 scripts/profile_summary.py --stats "$S" --fetch "$F" --write "$W" --steps 2 --tag "$TAG" --config "${PROFILE_CONFIG:-{\"samples\":50000,\"features\":3000,\"trees\":200}}"
python3 scripts/trace_launches.py "$T" --timeline > "profiles/${TAG}_step_timeline.txt" 2>/dev/null || true
mkdir -p "gpurun_out/profiles_$TAG"
cp profiles/${TAG}_* "gpurun_out/profiles_$TAG/"
cp "$OUT/bench_under_rocprof.json" "gpurun_out/profiles_$TAG/${TAG}_bench_under_rocprof.json"
head -30 "$S"
