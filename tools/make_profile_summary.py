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
            m = re.search(r"((?:fast|wr|mx)_(?:row|col)pass\d*_u8|mx_altsums_reduce|mx_altsums|mx_quirk_terms|fx_blur_u8|fx_prepass|fx_altsums|fx_edge_strips|fx_quirk_reduce|fx_quirk_cols)", k)
            name = m.group(1) if m else None
            if name:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, \
           {k: {c: len(v) for c, v in d.items()} for k, d in acc.items()}

summary = {}
# SQ counter summary (VALU / LDS busy, bank conflicts, wait and issue-stall cycles) beside the traffic counters
sq = {}
for sub in ("sq_a", "sq_b", "sq_c", "sq_d"):
    m, n = mean_counters(sub)
    for k in m:
        sq.setdefault(k, {}).update({c: round(v, 1) for c, v in m[k].items()})
for k, d in sq.items():
    # derived: SQ_*_CYCLES count quad-cycles summed over waves; busy fractions against the wave-cycles of the launch
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for name, c in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY"),
                        ("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("lds_active_frac", "SQ_ACTIVE_INST_LDS")):
            if c in d:
                d[name] = round(d[c] / wc, 4)
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_f16), SQ_WAVE_CYCLES quad-cycles summed over the waves;
        # with one wave per SIMD (the fused kernel) the quotient is the fraction of the kernel the matrix pipe is busy
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy_frac_one_wave_per_simd"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
if sq:
    json.dump(sq, open(os.path.join(out, "%s_sq_counters.json" % tag), "w"), indent=1, sort_keys=True)

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
    # wr_colpass_u8 (8-byte pieces of 24-byte strip rows) and wr_rowpass_u8 (8 bytes per lane, 64-byte records): calibrated
    # on their known input sizes -- FETCH_SIZE reads 99.97 MB for the 199.07 MB of u8 frames and 402.6 MB for the 796-800 MB
    # of float intermediate, i.e. one half in both cases, like the wide reads: f = 2
    f = 1.0 if k == "fast_rowpass_u8" else 2.0
    if not k.endswith("pass_u8") and not k.endswith("pass3_u8") and not k.startswith("fx_"):
        continue                                       # the quirk's small kernels: counters kept in *_pmc_counters.json only
    if "FETCH_SIZE" in summary[k] and "WRITE_SIZE" in summary[k]:
        fetch, write = summary[k]["FETCH_SIZE"] * 1024, summary[k]["WRITE_SIZE"] * 1024
        traffic[k] = {
            "frames_per_launch": frames_per_launch,
            "fetch_size_raw_bytes": fetch, "fetch_correction": f, "write_size_bytes": write,
            "hbm_bytes_per_launch": fetch * f + write,
            # two-pass kernels: 15 B/px each; the fused kernel has to move 6 B/px (3 in + 3 out), its pre-pass 3 B/px
            "alg_bytes_per_launch": ({"fx_blur_u8": 6, "fx_altsums": 3, "fx_prepass": 3}.get(k, 15) * px * frames_per_launch) if not k.startswith("fx_") or k in ("fx_blur_u8", "fx_altsums", "fx_prepass") else None,
            "l2_hit_rate": round(summary[k]["TCC_HIT_sum"] / (summary[k]["TCC_HIT_sum"] + summary[k]["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in summary[k] else None,
            "source": "%s_pmc_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate runs of bench.py)" % tag,
        }
# keep the entries of kernel families this run did not execute (the FFT kernels when the bench ran the matrix-core engine, and back)
try:
    merged = json.load(open(os.path.join(out, "pmc_traffic.json")))
except Exception:
    merged = {}
merged.update(traffic)
traffic = merged
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
