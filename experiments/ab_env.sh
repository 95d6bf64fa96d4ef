#!/bin/bash
# Same-box A/B of the headline bench under two environments.  Usage: ab_env.sh "VAR=a" "VAR=b"
for round in 1 2; do
  for v in "$1" "$2"; do
    env $v timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-sweep --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'tok/s',d['value'],'ms/step',d['ms_per_step'],'AR',d['ar_tokens_per_s'],'speedup',d['speedup_vs_ar'],'exact',d['spec_stream_equals_ar_stream'],'draft pass us',d['draft_pass_roofline']['us_per_pass'])" || exit 1
  done
done
