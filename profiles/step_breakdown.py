"""Breaks one speculative step out of a rocprofv3 kernel trace CSV:
   python profiles/step_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'advance_state' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
seg = rows[a + 1:b + 1]
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print(f"kernels in step: {len(seg)}  wall us: {(t1 - t0) / 1e3:.1f}  busy us: {busy / 1e3:.1f}")
agg = collections.OrderedDict()
for r in seg:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)[:52]
    key = (n, r['Grid_Size_X'] + '/lds' + r['LDS_Block_Size'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = agg.setdefault(key, [0, 0.0])
    g[0] += 1
    g[1] += d
for (n, gx), (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n:52s} grid={gx:>16s} n={c:4d} total_us={t:9.1f} avg={t / c:7.2f}")
