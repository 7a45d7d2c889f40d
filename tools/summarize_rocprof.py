"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a small, committed summary: one table per kernel NAME
(template arguments stripped - the names bench.py's roofline uses) and one per instantiation."""
import collections
import csv
import re
import sys


def base_name(name: str) -> str:
    n = name.replace("void ", "").replace("s2k::", "")
    return re.split(r"[<(]", n)[0]


def main(src, dst, title):
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    agg = collections.OrderedDict()
    for r in rows:
        e = agg.setdefault(base_name(r["Name"]), [0, 0.0])
        e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"])
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: rocprofv3 --kernel-trace --stats (kernel_stats.csv); total kernel time {tot/1e6:.2f} ms\n\n")
        f.write("## by kernel name (all template instantiations together)\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for name, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"| `{name}` | {calls} | {ns/1e6:.3f} | {ns/calls/1e3:.1f} | {100*ns/tot:.2f} |\n")
        f.write("\n## by instantiation\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            name = r["Name"]
            if len(name) > 110:
                name = name[:107] + "..."
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
