#!/bin/bash
# Ad-hoc counter sweep: bash tools/pmc_sweep.sh <workload> <counter> [<counter> ...] -- one rocprofv3 pass per counter,
# per-kernel sums printed by tools/pmc_sum.py.  Output under gpurun_out/pmc_sweep/.
set -u
WL=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_sweep
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in "$@"; do
    rm -rf "$OUT/raw_$c"
    timeout -k 10 300 rocprofv3 --pmc $c -d "$OUT/raw_$c" -f csv -- python3 "$R/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$c.log" 2>&1
    echo "pass $c rc=$?"
    f=$(find "$OUT/raw_$c" -name '*counter_collection.csv' 2>/dev/null | head -1)
    [ -n "$f" ] && python3 "$R/tools/pmc_sum.py" "$f" > "$OUT/$c.txt" && rm -rf "$OUT/raw_$c"
done
