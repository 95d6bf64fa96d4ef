#!/bin/bash
# Same-box sweep of the headline bench over several sets of bench.py options (one line each).
# Usage: experiments/sweep_bench.sh "<options 1>" "<options 2>" ...
for v in "$@"; do
  timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-sweep --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'tok/s',d['value'],'ms/step',d['ms_per_step'],'AR',d['ar_tokens_per_s'],'speedup',d['speedup_vs_ar'],'exact',d['spec_stream_equals_ar_stream'],'draft pass us',d['draft_pass_roofline']['us_per_pass'])" || exit 1
done
