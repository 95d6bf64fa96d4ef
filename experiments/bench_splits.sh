#!/bin/bash
# Headline bench at several attention split sizes (target / draft).
for cfg in "256 0" "256 2048" "128 2048" "512 2048" "256 1024"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 48 --warmup 8 --no-sweep --no-cpu-baseline --split-tokens $1 --draft-split-tokens $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('split',$1,'draft',$2,'tok/s',d['value'],'ms/step',d['ms_per_step'],'AR',d['ar_tokens_per_s'],'speedup',d['speedup_vs_ar'],'exact',d['spec_stream_equals_ar_stream'])" || exit 1
done
