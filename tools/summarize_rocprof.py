"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a small, committed summary."""
import csv
import sys


def main(src, dst, title):
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: rocprofv3 --kernel-trace --stats (kernel_stats.csv); total kernel time {tot/1e6:.2f} ms\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            name = r["Name"]
            if len(name) > 110:
                name = name[:107] + "..."
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
