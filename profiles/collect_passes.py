"""Per-kernel HBM traffic of the draft's decode pass and the target's verify pass, from the raw
rocprofv3 output that experiments/profile_passes.sh leaves under gpurun_out/<round>_profile:

  <round>_pmc_hbm.json   for every launch position of a layer (QKV, attention, [merge,] output
                         projection, gate/up, down) and the LM head: kernel, launches averaged,
                         FETCH_SIZE / WRITE_SIZE, HBM bytes read and written per launch, the bytes
                         the launch must move, their ratio, and the launch's duration from the
                         kernel trace of the same script

usage: python profiles/collect_passes.py [round-prefix, default r3]
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("linear_kernel", "paged_attn_kernel", "attn_merge_kernel", "embed_kernel")


def short(name):
    return name.replace("(anonymous namespace)::", "")


def dispatches(path, value_of):
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            v = value_of(row)
            if v is not None:
                rows.append((int(row["Dispatch_Id"]), short(row["Kernel_Name"]), v))
    rows.sort()
    return rows


def passes(rows):
    """[(kernel, value)] of the launches between the markers: draft pass, target pass."""
    out, cur = [], None
    for _, name, v in rows:
        if "fill_uniforms" in name:
            if cur is not None:
                out.append(cur)
            cur = []
        elif cur is not None and any(k in name for k in OURS):
            cur.append((name, v))
    return out


def label_pass(launches):
    """Position labels: a layer starts at the rotary QKV launch (epilogue 100)."""
    starts = [i for i, (n, _) in enumerate(launches) if "linear_kernel<100" in n]
    period = starts[1] - starts[0]
    names = {5: ["qkv", "attention", "o_proj", "gate_up", "down"],
             6: ["qkv", "attention", "merge", "o_proj", "gate_up", "down"]}[period]
    labelled = []
    for i, (n, v) in enumerate(launches):
        if i < starts[0]:
            lab = "embed"
        elif i >= starts[-1] + period:
            lab = "lm_head"
        else:
            lab = names[(i - starts[0]) % period]
        labelled.append((lab, n, v))
    return labelled


def averages(labelled):
    acc = {}
    for lab, n, v in labelled:
        e = acc.setdefault(lab, {"kernel": n, "vals": []})
        e["vals"].append(v)
    return {lab: (e["kernel"], sum(e["vals"]) / len(e["vals"]), len(e["vals"])) for lab, e in acc.items()}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
    src = os.path.join(ROOT, "gpurun_out", f"{rnd}_profile")
    counter = lambda c: (lambda row: float(row["Counter_Value"]) if row["Counter_Name"] == c else None)
    fetch = passes(dispatches(os.path.join(src, "passes_FETCH_SIZE", "pmc_counter_collection.csv"),
                              counter("FETCH_SIZE")))
    write = passes(dispatches(os.path.join(src, "passes_WRITE_SIZE", "pmc_counter_collection.csv"),
                              counter("WRITE_SIZE")))
    trace_path = os.path.join(src, "passes_trace", "k_kernel_trace.csv")
    dur = None
    if os.path.exists(trace_path):
        dur = passes(dispatches(trace_path, lambda row: (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    alg = {}
    with open(os.path.join(src, "passes_FETCH_SIZE.log")) as f:
        for line in f:
            m = re.match(r"(\S+) rows (\d+) layers (\d+) algorithmic bytes per launch: (\{.*\}) lm_head (\d+)", line)
            if m:
                d = json.loads(m.group(4).replace("'", '"'))
                d["lm_head"] = int(m.group(5))
                alg[len(alg)] = (m.group(1), int(m.group(2)), int(m.group(3)), d)
    out = {"unit_note": "rocprofv3 FETCH_SIZE / WRITE_SIZE in KiB per dispatch, one counter per run of "
                        "experiments/pmc_passes.py (eager passes, every layer its own weights, caches flushed "
                        "before each pass); gfx950 FETCH_SIZE counts 128-B read requests at 64 B, so HBM read "
                        "bytes = 2 x FETCH_SIZE x 1024; durations from a --kernel-trace run of the same script "
                        "(eager launches: a few us longer than inside the replayed graph)",
           "passes": {}}
    for i, which in enumerate(("draft_decode_pass", "target_verify_pass")):
        name, rows, layers, a = alg[i]
        f_avg, w_avg = averages(label_pass(fetch[i])), averages(label_pass(write[i]))
        d_avg = averages(label_pass(dur[i])) if dur else {}
        entry = {"model": name, "rows": rows, "layers": layers, "cached_tokens": 1089, "kernels": {}}
        for lab, (kern, favg, n) in f_avg.items():
            e = {"kernel": kern, "launches": n, "FETCH_SIZE_KiB_avg": round(favg, 2),
                 "WRITE_SIZE_KiB_avg": round(w_avg[lab][1], 2),
                 "hbm_read_bytes_per_launch": int(round(2 * favg * 1024)),
                 "hbm_write_bytes_per_launch": int(round(w_avg[lab][1] * 1024))}
            if lab in a:
                e["algorithmic_bytes_per_launch"] = a[lab]
                e["read_over_algorithmic"] = round(e["hbm_read_bytes_per_launch"] / a[lab], 3)
            if lab in d_avg:
                e["avg_us_eager"] = round(d_avg[lab][1], 2)
            entry["kernels"][lab] = e
        out["passes"][which] = entry
    hbm_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm.json")
    merged = {}
    if os.path.exists(hbm_path):          # keep what profiles/collect_round.py put there
        with open(hbm_path) as f:
            merged = json.load(f)
    merged["passes_unit_note"] = out["unit_note"]
    merged["passes"] = out["passes"]
    with open(hbm_path, "w") as f:
        json.dump(merged, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
