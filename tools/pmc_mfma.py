"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):
util = MFMA busy cycles summed over all SIMDs / (GRBM_GUI_ACTIVE x 32 CUs x 4 SIMDs).
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md) summed over the chip;
rocprofv3 reports GRBM_GUI_ACTIVE per dispatch summed over the 8 XCDs (2.6 M "cycles" for a 165 us launch), hence the
per-XCD normalisation by 32 CUs x 4 SIMDs.  Cross-check: conv3p 28 % here vs 27.6 % of the bf16 peak from HIP events."""
import collections, csv, json, re, sys
busy, act, n = collections.Counter(), collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()[:64]
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += v; n[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += v
out = {}
for k in busy:
    if busy[k] > 0 and act[k] > 0:
        out[k] = {"launches": n[k], "mfma_busy_cycles_per_launch": busy[k] / n[k], "active_cycles_per_launch": act[k] / n[k],
                  "mfma_util": busy[k] / (act[k] * 32 * 4)}
import os
res = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))
res["head"] = os.environ.get("UDT_HEAD", "unknown")       # the commit these counters were collected at
print(json.dumps(res, indent=1))
