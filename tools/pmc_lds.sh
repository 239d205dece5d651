#!/bin/bash
# LDS / MFMA counters of a dense config's kernels (run on the GPU box from the repo root):
#   bash tools/pmc_lds.sh <tag> <config> [dtype]   -> gpurun_out/pmclds_<tag>.txt
set -u
TAG=${1:-k}; CFG=${2:-unet_big}; DT=${3:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmclds_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT" -f csv -- python3 "$R/tools/profile_step.py" "$CFG" --dtype "$DT" --steps 2 > "$OUT/log.txt" 2>&1
echo "rc=$?"
python3 - "$OUT" > "$R/gpurun_out/pmclds_$TAG.txt" <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
names = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-50:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] not in names: names.append(r['Counter_Name'])
        if r['Counter_Name'] == names[0]: n[k] += 1
print('%-52s %5s ' % ('kernel', 'n') + ' '.join('%16s' % x[3:][:16] for x in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][names[0]] if names else 0):
    print('%-52s %5d ' % (k, n[k]) + ' '.join('%16.0f' % (v[x] / max(n[k], 1)) for x in names))
PY
head -30 "$R/gpurun_out/pmclds_$TAG.txt"
tail -5 "$OUT/log.txt"
