#!/bin/bash
# rocprofv3 kernel-trace statistics of one train-step loop (run on the GPU box from the repo root):
#   bash tools/kstats.sh <tag> [unet|mulmo|unet_big] [extra profile_step.py args]   -> gpurun_out/kstats_<tag>.csv
set -u
TAG=${1:-k}
CFG=${2:-unet}
shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstats_raw_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -f csv -- python3 "$R/tools/profile_step.py" "$CFG" --steps 10 "$@" > "$OUT/log.txt" 2>&1
echo "rc=$?"
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cp "$f" "$R/gpurun_out/kstats_$TAG.csv"
head -25 "$R/gpurun_out/kstats_$TAG.csv" | cut -c1-200
