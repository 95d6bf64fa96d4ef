"""Breaks one speculative step out of a rocprofv3 kernel trace (CSV or rocpd .db):
   python profiles/step_breakdown.py <kernel_trace.csv | results.db> [steps_back]

Per kernel kind (name, grid, LDS): launches in the step, the slot time (start of
the kernel to start of the next one, i.e. duration + boundary) and the duration."""
import collections
import csv
import re
import sqlite3
import sys


def load(path):
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        q = "select name, start, end, grid_x, grid_y, grid_z, lds_size from kernels order by start"
        return [dict(name=n, start=s, end=e, grid=f"{gx}x{gy}x{gz}", lds=l)
                for n, s, e, gx, gy, gz, l in db.execute(q)]
    rows = list(csv.DictReader(open(path)))
    out = [dict(name=r["Kernel_Name"], start=int(r["Start_Timestamp"]), end=int(r["End_Timestamp"]),
                grid=f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}",
                lds=r["LDS_Block_Size"]) for r in rows]
    out.sort(key=lambda r: r["start"])
    return out


def main():
    rows = load(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    idx = [i for i, r in enumerate(rows) if "advance_state" in r["name"] or "verify_from_partials" in r["name"]]
    a, b = idx[-1 - back], idx[-back]
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0]["start"], seg[-1]["end"]
    busy = sum(r["end"] - r["start"] for r in seg)
    print(f"kernels in step: {len(seg)}  wall us: {(t1 - t0) / 1e3:.1f}  busy us: {busy / 1e3:.1f}")
    agg = collections.OrderedDict()
    for i, r in enumerate(seg):
        n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
        n = re.sub(r"^void ", "", n)[:50]
        key = (n, f"{r['grid']}/lds{r['lds']}")
        nxt = seg[i + 1]["start"] if i + 1 < len(seg) else r["end"]
        g = agg.setdefault(key, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += (r["end"] - r["start"]) / 1e3
        g[2] += (nxt - r["start"]) / 1e3
    for (n, gx), (c, t, sl) in sorted(agg.items(), key=lambda x: -x[1][2]):
        print(f"{n:50s} {gx:>22s} n={c:4d} slot_us={sl:8.1f} avg_slot={sl / c:7.2f} avg_dur={t / c:7.2f}")


if __name__ == "__main__":
    main()
