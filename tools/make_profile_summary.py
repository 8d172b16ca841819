#!/usr/bin/env python3
"""tools/make_profile_summary.py gpurun_out/prof_TAG rNN -- copies the rocprofv3 summaries the bench
numbers are judged against into profiles/ (tracked) and derives profiles/pmc_traffic.json.

HBM traffic per launch = FETCH_SIZE * f + WRITE_SIZE (rocprofv3 reports KB), collected in SEPARATE
--pmc passes (MI355X_MICROARCH.md "rocprofv3 PMC slots").  gfx950 correction (same guide, "HBM"):
FETCH_SIZE tallies 128-byte requests at 64 bytes, i.e. reads exactly half the bytes of wide line
fills, so f = 2 for the column pass (its 16-byte-per-lane strip gather is served by 128-byte line
fills: with f = 2 the read side lands on the known byte count of the float planes) and for
fast_rowpass3_u8 (whole u8 rows by 16-byte loads).  fast_rowpass_u8 reads u8 with byte loads; its
FETCH_SIZE is within 11 % of the known input bytes with f = 1 and is reported uncorrected.
WRITE_SIZE is exact for streaming stores."""
import csv, glob, json, os, re, shutil, sys, collections

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

def newest(pattern):
    """gpurun merges every call's files into the same directory: only the latest run counts"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []

for f in newest(os.path.join(src, "stats", "**", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, "%s_kernel_stats.csv" % tag))
for f in glob.glob(os.path.join(src, "bench_under_rocprof.log")):
    shutil.copy(f, os.path.join(out, "%s_bench_under_rocprof.log" % tag))

def mean_counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(os.path.join(src, sub, "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            m = re.search(r"(fast_(?:row|col)pass\d*_u8)", k)
            name = m.group(1) if m else None
            if name:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, \
           {k: {c: len(v) for c, v in d.items()} for k, d in acc.items()}

summary = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
    m, n = mean_counters(sub)
    for k in m:
        summary.setdefault(k, {}).update({c: round(v, 1) for c, v in m[k].items()})
        summary[k]["dispatches_" + sub] = max(n[k].values())
json.dump(summary, open(os.path.join(out, "%s_pmc_counters.json" % tag), "w"), indent=1, sort_keys=True)

frames_per_launch = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
px = 2160 * 3840
traffic = {}
for k in sorted(summary):
    # wide (16-byte per lane, whole-line) reads are the ones FETCH_SIZE halves: the column gather and the staged
    # row input of fast_rowpass3_u8; the byte loads of fast_rowpass_u8 are reported uncorrected
    f = 1.0 if k == "fast_rowpass_u8" else 2.0
    if "FETCH_SIZE" in summary[k] and "WRITE_SIZE" in summary[k]:
        fetch, write = summary[k]["FETCH_SIZE"] * 1024, summary[k]["WRITE_SIZE"] * 1024
        traffic[k] = {
            "frames_per_launch": frames_per_launch,
            "fetch_size_raw_bytes": fetch, "fetch_correction": f, "write_size_bytes": write,
            "hbm_bytes_per_launch": fetch * f + write,
            "alg_bytes_per_launch": 15 * px * frames_per_launch,
            "l2_hit_rate": round(summary[k]["TCC_HIT_sum"] / (summary[k]["TCC_HIT_sum"] + summary[k]["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in summary[k] else None,
            "source": "%s_pmc_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate runs of bench.py)" % tag,
        }
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
