#!/bin/bash
# SQ wave-cycle breakdown of the unet.yaml step's kernels (run on the GPU box from the repo root):
#   bash tools/pmc_kernel.sh <tag> [unet|mulmo|unet_big] [extra profile_step.py args]
#       -> gpurun_out/pmc_<tag>.txt (per kernel: counters averaged over the launches of 3 steps)
set -u
TAG=${1:-k}
CFG=${2:-unet}
shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d "$OUT" -f csv -- python3 "$R/tools/profile_step.py" "$CFG" --steps 3 "$@" > "$OUT/log.txt" 2>&1
echo "rc=$?"
python3 - "$OUT" > "$R/gpurun_out/pmc_$TAG.txt" <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-60:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVES"]
print('%-62s %5s ' % ('kernel', 'n') + ' '.join('%14s' % x[3:] for x in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES']):
    print('%-62s %5d ' % (k, n[k]) + ' '.join('%14.0f' % (v[x] / max(n[k], 1)) for x in names))
PY
cat "$R/gpurun_out/pmc_$TAG.txt" | head -40
