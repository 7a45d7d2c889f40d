"""GPU busy time from a rocprofv3 --kernel-trace CSV: union of the kernel intervals (all streams) against the wall span,
over the last `--steps` repetitions of a marker kernel (default: the fused Adam step, one launch per training step).
    python tools/busy_union.py <kernel_trace.csv> [--marker adam_kernel] [--steps 10]"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="adam")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip-last", type=int, default=0, help="marker launches after the timed region (bench.py times Adam alone 11 times)")
    ap.add_argument("--per-step", type=int, default=1, help="marker launches per step (FlatAdam: one per contiguous parameter range)")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if a.skip_last:
        marks = marks[:-a.skip_last * a.per_step]
    marks = marks[a.per_step - 1::a.per_step]      # the last marker launch of every step
    if len(marks) < a.steps + 1:
        raise SystemExit(f"only {len(marks)} marker launches")
    lo, hi = marks[-a.steps - 1], marks[-1]
    t0, t1 = rows[lo][1], rows[hi][1]              # from the end of one Adam to the end of the last
    busy, cur_s, cur_e, sum_all = 0, None, None, 0
    for s, e, _ in rows[lo + 1:hi + 1]:
        sum_all += e - s
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    n = a.steps
    print(f"{n} steps: wall {(t1 - t0) / n / 1e6:.3f} ms/step, GPU busy (union of kernels) {busy / n / 1e6:.3f} ms/step "
          f"({100.0 * busy / (t1 - t0):.1f} %), idle {(t1 - t0 - busy) / n / 1e6:.3f} ms/step, sum of kernel durations {sum_all / n / 1e6:.3f} ms/step "
          f"(overlap of the two streams {(sum_all - busy) / n / 1e6:.3f} ms/step), {(hi - lo) / n:.0f} launches/step")


if __name__ == "__main__":
    main()
