#!/bin/bash
# Per-kernel statistics of one model's decode pass: experiments/profile_draft_pass.sh <model> <ctx> <q_len> [B]
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pass_profile
mkdir -p $OUT
export SD_B=${4:-1}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- \
  python3 $ROOT/experiments/draft_pass.py $1 $2 $3 > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} n={r["Calls"]:>5s} avg={float(r["AverageNs"])/1e3:7.2f} us  {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
