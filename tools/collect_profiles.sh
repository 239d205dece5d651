#!/bin/bash
# Collects the round's rocprofv3 evidence on a GPU box (run from the repo root: bash tools/collect_profiles.sh [unet|dense|all]).
# One counter (or tracing option) per pass, the profiled program directly after `--`; summaries land under gpurun_out/prof/
# and are copied into profiles/ by hand afterwards (see profiles/README.md).
set -u
WHAT=${1:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

pass() {   # pass <name> <rocprofv3 options...> -- <program...>
    local name=$1; shift
    rm -rf "$OUT/raw_$name"
    timeout -k 10 600 rocprofv3 "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc" | tee -a "$OUT/passes.txt"
    for f in $(find "$OUT/raw_$name" -name '*.csv' 2>/dev/null); do
        cp "$f" "$OUT/${name}_$(basename "$f" | sed 's/^[0-9]*_//')"
    done
    return $rc
}

if [ "$WHAT" = unet ] || [ "$WHAT" = all ]; then
    pass unet_stats --kernel-trace --stats -d "$OUT/raw_unet_stats" -f csv -- python3 "$R/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-other-workloads || exit 1
    pass unet_fetch --pmc FETCH_SIZE -d "$OUT/raw_unet_fetch" -f csv -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-other-workloads || exit 1
    pass unet_write --pmc WRITE_SIZE -d "$OUT/raw_unet_write" -f csv -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-other-workloads || exit 1
fi
if [ "$WHAT" = dense ] || [ "$WHAT" = all ]; then
    for wl in unet_big mulmo_unet; do
        pass ${wl}_stats --kernel-trace --stats -d "$OUT/raw_${wl}_stats" -f csv -- python3 "$R/bench.py" --workload $wl --steps 20 --warmup 5 --no-cpu-baseline || exit 1
        pass ${wl}_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/raw_${wl}_mfma" -f csv -- python3 "$R/bench.py" --workload $wl --steps 3 --warmup 2 --no-cpu-baseline || exit 1
        pass ${wl}_busy --pmc SQ_BUSY_CU_CYCLES -d "$OUT/raw_${wl}_busy" -f csv -- python3 "$R/bench.py" --workload $wl --steps 3 --warmup 2 --no-cpu-baseline || exit 1
    done
fi
ls -la "$OUT" | head -60
