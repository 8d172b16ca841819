"""tools/fx_gaps.py ASM [KERNEL-SUBSTRING] -- static look at the MFMA stream of a kernel: instructions between consecutive MFMAs.
Cost model (one wave per SIMD, MI355X guide): an MFMA occupies the matrix pipe for 32 cycles and the issue port for 8; a vector
instruction 4 (transcendental 8), DS / VMEM issue 4, s_nop N: N + 1; waits are not modelled.  Prints the distribution of gap costs and
the estimated cycles per loop body = sum over gaps of max(32, 8 + cost)."""
import re
import sys

asm = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2] if len(sys.argv) > 2 else "fx_blur_u8ILi11ELb0E"
start = next(i for i, l in enumerate(asm) if l.startswith("_ZN") and sub in l)
end = next(i for i in range(start, len(asm)) if asm[i].strip().startswith("s_endpgm"))
body = asm[start:end]
# the main loop: from the first s_barrier after the prologue's second one to the last backward branch
gaps, cur, kinds = [], 0, {}
in_loop = False
nb = 0
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        nb += 1
    if nb < 2:
        continue
    if op.startswith("v_mfma"):
        gaps.append(cur)
        cur = 0
        continue
    if op.startswith("s_nop"):
        c = int(t.split()[1]) + 1
    elif op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log"):
        c = 8
    elif op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_"):
        c = 4
    elif op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        c = 0
    else:
        c = 1
    cur += c
    k = op.split("_")[0] + "_" + (op.split("_")[1] if "_" in op else "")
    kinds[k] = kinds.get(k, 0) + 1
est = sum(max(32, 8 + g) for g in gaps)
print("MFMAs after the prologue: %d, estimated cycles %d (%.1f per MFMA), ideal %d" % (len(gaps), est, est / max(len(gaps), 1), 32 * len(gaps)))
hist = {}
for g in gaps:
    b = min(g // 8 * 8, 200)
    hist[b] = hist.get(b, 0) + 1
print("gap cost histogram (cycles of non-MFMA issue between two MFMAs):")
for b in sorted(hist):
    print("  %3d..%3d: %4d  extra %6d" % (b, b + 7, hist[b], sum(max(0, 8 + g - 32) for g in gaps if min(g // 8 * 8, 200) == b)))
print("instruction kinds:", sorted(kinds.items(), key=lambda kv: -kv[1])[:16])
