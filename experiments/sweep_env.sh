#!/bin/bash
# Same-box sweep: each argument is "ENV=VAL ... -- <bench options>".
for v in "$@"; do
  envs="${v%%--*}"; opts="${v#*-- }"
  env $envs timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-sweep --no-cpu-baseline $opts 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'tok/s',d['value'],'ms/step',d['ms_per_step'],'AR',d['ar_tokens_per_s'],'speedup',d['speedup_vs_ar'],'exact',d['spec_stream_equals_ar_stream'],'draft pass us',d['draft_pass_roofline']['us_per_pass'])" || exit 1
done
