"""Per-step kernel-time breakdown from a rocprofv3 --kernel-trace CSV.

Steps are delimited by the optimizer kernel (one launch per training step); warm-up steps,
which contain the convolution library's one-off algorithm search, are excluded.

usage: python tools/summarize_trace.py <trace_kernel_trace.csv> [--steps 3] [--delim sgdm_kernel] [--top 40]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--delim", default="sgdm_kernel")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [e for s, e, n in rows if a.delim in n]
    if len(ends) < a.steps + 1:
        raise SystemExit(f"only {len(ends)} '{a.delim}' launches in the trace")
    t0, t1 = ends[-a.steps - 1], ends[-1]
    sel = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in sel:
        agg[short(n)][0] += e - s
        agg[short(n)][1] += 1
        busy += e - s
    wall = (t1 - t0) / a.steps / 1e6
    print(f"steps analysed: {a.steps}; wall per step {wall:.2f} ms; kernel time per step {busy / a.steps / 1e6:.2f} ms "
          f"({100 * busy / (t1 - t0):.1f}% of wall); kernels per step {len(sel) / a.steps:.0f}")
    print(f"{'pct':>6} {'ms/step':>9} {'calls/step':>10} {'avg_us':>9}  kernel")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print(f"{100 * t / busy:6.2f} {t / a.steps / 1e6:9.3f} {c / a.steps:10.1f} {t / c / 1e3:9.1f}  {n}")


if __name__ == "__main__":
    main()
