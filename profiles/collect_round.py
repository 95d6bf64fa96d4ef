"""Builds the committed per-round summaries from the raw rocprofv3 output that
experiments/profile_round.sh and experiments/profile_kernels.sh leave under
gpurun_out/<round>_profile:

  <round>_bench_kernel_stats.csv     per-kernel statistics of the headline bench
  <round>_roofline_kernel_stats.csv  statistics of the two roofline kernels run alone
  <round>_pmc_hbm.json               "kernels": HBM bytes per launch of the two roofline kernels from the
                                     FETCH_SIZE / WRITE_SIZE passes ("passes": see collect_passes.py)
  <round>_step_breakdown.txt         kernels of one speculative step, grouped
  <round>_pmc_mfma.json              matrix-pipe utilisation of the roofline kernels
                                     (experiments/profile_mfma.sh), when that pass was run

usage: python profiles/collect_round.py [round-prefix, default r2]
"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGORITHMIC = {  # bytes one launch must move (weights of one gate/up matrix; KV of 32 x 4096 tokens)
    "linear_kernel": ("algorithmic_bytes_per_launch", 235102208, "experiments/pmc_linear.py"),
    "paged_attn_kernel": ("algorithmic_kv_bytes_per_launch", 537657344, "experiments/pmc_attention.py"),
    "attn_merge_kernel": (None, None, "experiments/pmc_attention.py"),
}


def short(name):
    return name.replace("(anonymous namespace)::", "")


def counter_avg(path, counter):
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            per.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r2"
    src = os.path.join(ROOT, "gpurun_out", f"{rnd}_profile")
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"),
                os.path.join(dst, f"{rnd}_bench_kernel_stats.csv"))
    rows, header = [], None
    for sub in ("stats_linear", "stats_attn"):
        with open(os.path.join(src, sub, "k_kernel_stats.csv")) as f:
            r = csv.reader(f)
            header = next(r)
            rows += [x for x in r if any(k in x[0] for k in ALGORITHMIC)]
    with open(os.path.join(dst, f"{rnd}_roofline_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(header)
        w.writerows(rows)
    kernels = {}
    for tag, sub in (("linear", "pmc_linear"), ("attn", "pmc_attn")):
        fetch = counter_avg(os.path.join(src, f"{sub}_FETCH_SIZE", "pmc_counter_collection.csv"), "FETCH_SIZE")
        write = counter_avg(os.path.join(src, f"{sub}_WRITE_SIZE", "pmc_counter_collection.csv"), "WRITE_SIZE")
        for name, (favg, n) in fetch.items():
            key = next((k for k in ALGORITHMIC if k in name), None)
            if key is None:
                continue
            field, alg, script = ALGORITHMIC[key]
            wavg = write.get(name, (0.0, 0))[0]
            entry = {"script": script, "launches": n, "FETCH_SIZE_KiB_avg": round(favg, 2),
                     "WRITE_SIZE_KiB_avg": round(wavg, 2),
                     "hbm_read_bytes_per_launch": int(round(2 * favg * 1024)),
                     "hbm_write_bytes_per_launch": int(round(wavg * 1024))}
            if field:
                entry[field] = alg
            kernels[name] = entry
    out = {"unit_note": "rocprofv3 FETCH_SIZE / WRITE_SIZE in KiB per dispatch; gfx950 FETCH_SIZE counts 128-B read "
                        "requests at 64 B, so HBM read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for 16-B stores",
           "kernels": kernels}
    hbm_path = os.path.join(dst, f"{rnd}_pmc_hbm.json")
    if os.path.exists(hbm_path):          # keep what profiles/collect_passes.py put there
        with open(hbm_path) as f:
            old = json.load(f)
        for key in ("passes", "passes_unit_note"):
            if key in old:
                out[key] = old[key]
    with open(hbm_path, "w") as f:
        json.dump(out, f, indent=1)
    mfma = {}
    for sub in ("pmc_linear_mfma", "pmc_attn_mfma"):
        path = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        names = ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                 "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES")
        vals = {n: counter_avg(path, n) for n in names}
        for name in vals["GRBM_GUI_ACTIVE"]:
            if not any(k in name for k in ALGORITHMIC):
                continue
            e = {n: round(vals[n][name][0]) for n in names if name in vals[n]}
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ counters over all 1024 SIMDs
            active = e["GRBM_GUI_ACTIVE"] / 8
            e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * 1024), 4)
            e["waves_waiting_frac"] = round(e["SQ_WAIT_ANY"] / max(e["SQ_WAVE_CYCLES"], 1), 3)
            mfma[name] = e
    if mfma:
        with open(os.path.join(dst, f"{rnd}_pmc_mfma.json"), "w") as f:
            json.dump({"note": "rocprofv3 --pmc, one pass per script; mfma_util = matrix-pipe busy cycles "
                               "/ (GPU-active cycles x 1024 SIMDs): these kernels use MFMA to avoid "
                               "cross-lane work, not for FLOPs, and are bound by HBM",
                       "kernels": mfma}, f, indent=1)
    trace = os.path.join(src, "stats", "bench_kernel_trace.csv")
    txt = subprocess.run([sys.executable, os.path.join(dst, "step_breakdown.py"), trace],
                         capture_output=True, text=True)
    if txt.returncode == 0:
        with open(os.path.join(dst, f"{rnd}_step_breakdown.txt"), "w") as f:
            f.write(txt.stdout)
    else:
        print(txt.stderr, file=sys.stderr)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
