"""Per-kernel means of the counters in a `rocprofv3 --pmc ... --output-format csv` directory.

    python tools/summarize_pmc.py gpurun_out/pmc_sq [gpurun_out/pmc_tcc ...] > profiles/pmc_summary.txt
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:60]


def main():
    for d in sys.argv[1:]:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print(f"# {d}: no counter_collection.csv")
            continue
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get("Kernel_Name", "?"))
                    if k.startswith("at::") or k.startswith("Cijk") or "miopen" in k.lower() or "rocprim" in k:
                        continue
                    k = f"{k} grid={row.get('Grid_Size', '?')}"
                    c = acc[k][row["Counter_Name"]]
                    c[0] += float(row["Counter_Value"])
                    c[1] += 1
        counters = sorted({c for k in acc for c in acc[k]})
        print(f"# {d}  (mean per dispatch)")
        derived = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters
        print("kernel".ljust(90) + "".join(c[-22:].rjust(24) for c in counters)
              + ("   mfma_busy_%".rjust(16) if derived else "") + "   dispatches")
        for k in sorted(acc):
            n = max(v[1] for v in acc[k].values())
            mean = {c: acc[k][c][0] / max(1, acc[k][c][1]) for c in counters if c in acc[k]}
            line = k.ljust(90) + "".join(("%.4g" % mean.get(c, float("nan"))).rjust(24) for c in counters)
            if derived:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs issue MFMAs
                wall = mean["GRBM_GUI_ACTIVE"] / 8.0
                line += ("%.1f" % (100.0 * mean["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, wall * 1024))).rjust(16)
            print(line + f"   {n}")
        print()


if __name__ == "__main__":
    main()
