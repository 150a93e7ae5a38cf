#!/bin/bash
# Round-3 profiles on the GPU box, everything from HEAD's library:
#   gpurun -- 'bash scripts/profile_r03.sh'     then copy gpurun_out/profiles_r03/* into profiles/
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
bash scripts/profile_c3.sh r03_c3 > gpurun_out/profile_c3.log 2>&1
OUT=$REPO/gpurun_out/prof_r03_c5
mkdir -p "$OUT" gpurun_out/profiles_r03
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python3 "$REPO/scripts/profile_c5.py" > "$REPO/gpurun_out/profiles_r03/r03_c5_exact.json" 2> "$OUT/err.log" )
cp "$(find "$OUT" -name '*kernel_stats.csv' | head -1)" gpurun_out/profiles_r03/r03_c5_exact_kernel_stats.csv
OUT=$REPO/gpurun_out/prof_r03_sq
mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o p -- python3 "$REPO/scripts/bench_single_query.py" 1 8 63 1 > "$OUT/sq.log" 2>&1 )
{ grep "^nq" "$OUT/sq.log"; python3 scripts/bench_single_query.py --trace "$(find "$OUT" -name '*kernel_trace.csv' | head -1)"; } > gpurun_out/profiles_r03/r03_single_query_timeline.txt
python3 scripts/bench_e2e_index.py --out gpurun_out/profiles_r03/r03_e2e_index.json > /dev/null 2> gpurun_out/e2e.err
cp gpurun_out/profiles_r03_c3/* gpurun_out/profiles_r03/ 2>/dev/null || true
ls gpurun_out/profiles_r03
