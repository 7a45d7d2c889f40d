"""Per-stream view of a rocprofv3 --kernel-trace CSV over the last `--steps` training steps (marker = the fused Adam step):
sum of kernel durations per stream / queue, union busy time per stream, and the time only ONE stream is busy."""
import argparse
import collections
import csv


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="adam")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip-last", type=int, default=0)
    ap.add_argument("--per-step", type=int, default=1)
    ap.add_argument("--gap-us", type=float, default=20.0)
    ap.add_argument("--tail-us", type=float, default=0.0, help="print the kernel timeline of this many us before the last long main-queue gap ends")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if a.skip_last:
        marks = marks[:-a.skip_last * a.per_step]
    marks = marks[a.per_step - 1::a.per_step]
    lo, hi = marks[-a.steps - 1], marks[-1]
    t0, t1 = rows[lo][1], rows[hi][1]
    sel = rows[lo + 1:hi + 1]
    n = a.steps
    by = collections.defaultdict(list)
    for s, e, k, q in sel:
        by[q].append((s, e))
    print(f"wall {(t1 - t0) / n / 1e6:.3f} ms/step; union of all kernels {union([(s, e) for s, e, _, _ in sel]) / n / 1e6:.3f} ms/step")
    for q, iv in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(f"  queue {q}: {len(iv) / n:.0f} launches/step, sum {sum(e - s for s, e in iv) / n / 1e6:.3f} ms/step, union {union(iv) / n / 1e6:.3f} ms/step")
    # top kernels of the busiest queue by time
    main_q = max(by, key=lambda q: len(by[q]))
    agg = collections.defaultdict(float)
    for s, e, k, q in sel:
        if q == main_q:
            agg[k.split("(")[0].split("<")[0].replace("void ", "").replace("s2k::", "")] += e - s
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]:
        print(f"     main queue {k:32s} {v / n / 1e6:7.3f} ms/step")
    overlap_report(sel, main_q, n, a.gap_us, a.tail_us)


def overlap_report(sel, main_q, n, gap_us, tail_us=0.0):
    """time with only the main queue busy / only other queues / both / neither, and the main queue's long gaps (waits at joins)."""
    ev = []
    for s, e, _, q in sel:
        m = 0 if q == main_q else 1
        ev.append((s, 1, m))
        ev.append((e, -1, m))
    ev.sort()
    cnt = [0, 0]
    acc = collections.defaultdict(int)
    last = ev[0][0]
    for t, d, m in ev:
        acc[(cnt[0] > 0, cnt[1] > 0)] += t - last
        last = t
        cnt[m] += d
    print("  main only %.3f  side only %.3f  both %.3f  idle %.3f ms/step" % tuple(acc[k] / n / 1e6 for k in ((True, False), (False, True), (True, True), (False, False))))
    mq = [(s, e, k) for s, e, k, q in sel if q == main_q]
    side = sorted((s, e) for s, e, _, q in sel if q != main_q)
    gaps = []
    for (s0, e0, k0), (s1, e1, k1) in zip(mq, mq[1:]):
        if s1 - e0 > gap_us * 1000:
            busy = sum(max(0, min(e, s1) - max(s, e0)) for s, e in side if e > e0 and s < s1)
            gaps.append((s1 - e0, busy, k0, k1))
    tot = sum(g[0] for g in gaps)
    print(f"  main-queue gaps > {gap_us} us: {len(gaps) / n:.1f} per step, {tot / n / 1e6:.3f} ms/step (side busy during {sum(g[1] for g in gaps) / n / 1e6:.3f})")
    short = lambda k: k.split("(")[0].split("<")[0].replace("void ", "").replace("s2k::", "")[:28]
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"     gap {g[0] / 1e3:8.1f} us (side busy {g[1] / 1e3:7.1f})  after {short(g[2]):28s} before {short(g[3])}")
    if gaps and tail_us > 0:
        # timeline around the LAST long gap: what the side queue was finishing while the main queue waited
        g_end = max(s1 for (s0, e0, k0), (s1, e1, k1) in zip(mq, mq[1:]) if s1 - e0 > gap_us * 1000)
        lo = g_end - int(tail_us * 1000)
        print(f"  timeline of the last {tail_us:.0f} us before the main queue resumes (t relative to that moment, us):")
        for s_, e_, k, q in sel:
            if e_ > lo and s_ < g_end + 20000:
                print(f"     {'main' if q == main_q else 'side'} {(s_ - g_end) / 1e3:9.1f} .. {(e_ - g_end) / 1e3:9.1f}  {(e_ - s_) / 1e3:7.1f} us  {short(k)}")


if __name__ == "__main__":
    main()
