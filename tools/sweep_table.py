"""tools/sweep_table.py SWEEP.json [NOTE] -- the per-size table of `python bench.py --preset reference-sweep` (profiles/rNN_reference_sweep.txt)"""
import json
import sys
r = json.load(open(sys.argv[1]))
ok = [x for x in r["sizes"] if "ms" in x]
v = sorted(x["megapixels_per_s"] for x in ok)
print("min %.1f median %.1f max %.1f all_agree %s   (python bench.py --preset reference-sweep, one image per call%s)" % (
    v[0], v[len(v) // 2], v[-1], r["all_agree"], "; " + sys.argv[2] if len(sys.argv) > 2 else ""))
for x in r["sizes"]:
    if "ms" not in x:
        print("  %5dx%-5d %s" % (x["rows"], x["cols"], x.get("error")))
        continue
    print("  %5dx%-5d s=%6.1f N1/N0 %5d/%-5d %8.3f ms %9.0f MP/s  %-34s diff %d %.1e%s" % (
        x["rows"], x["cols"], x["sigma"], x["N1"], x["N0"], x["ms"], x["megapixels_per_s"], x.get("kernels", "?"), x["max_abs_diff_vs_generic"], x["frac_diff_vs_generic"],
        "   (run-time-planned kernels: %.3f ms)" % x["generic_ms"] if "generic_ms" in x else ""))
