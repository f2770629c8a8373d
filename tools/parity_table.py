"""gpurun_out/parity_per_yaml.jsonl (written by tests/test_gpu_configs.py on the MI355X) -> a markdown table of the MEASURED
parity of every in-scope YAML: max |delta| against the reference's PyTorch-CPU run, the tolerance asserted, label agreement and
the top-1 / top-2 margins of the flipped points.  usage: python tools/parity_table.py gpurun_out/parity_per_yaml.jsonl > profiles/rNN_parity_per_yaml.md"""
import json
import sys

rows = {}
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    rows[r["name"]] = r            # (the last run of a YAML wins)
print("# Measured parity per in-scope YAML (MI355X vs the reference's PyTorch-CPU goldens, tests/test_gpu_configs.py)\n")
print("| yaml | max abs delta | asserted tol | delta / tol | max abs ref | label agreement | flipped: margins top1 - top2 (ours) | boxes ref / gpu (unmatched) |")
print("|---|---|---|---|---|---|---|---|")
for name in sorted(rows):
    r = rows[name]
    fm = r.get("flipped_margins") or []
    box = "" if r["family"] != "pointpillars" else "%d / %d (%d, %d; budget %d)" % (
        r["boxes_ref"], r["boxes_gpu"], r["unmatched_ref_in_gpu"], r["unmatched_gpu_in_ref"], r["budget"])
    print("| %s | %.3g | %.3g | %.2f | %s | %s | %s | %s |" % (
        name, r["max_abs_delta"], r["tol"], r["max_abs_delta"] / r["tol"],
        "%.3g" % r["ref_abs_max"] if r.get("ref_abs_max") is not None else "",
        "%.6f (%d pts)" % (r["label_agreement"], r["points"]) if "label_agreement" in r else "",
        ", ".join("%.2g" % x for x in fm[:8]) + (" ..." if len(fm) > 8 else "") if fm else ("none" if "label_agreement" in r else ""), box))
