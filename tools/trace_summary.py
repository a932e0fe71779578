"""Per-step summary of a rocprofv3 --kernel-trace of bench.py: kernel time by name, launches, and the idle time BETWEEN
kernels inside the graph-replayed steps (steps are delimited by advance_iter_kernel).

    python tools/trace_summary.py <dir with *_kernel_trace.csv> [out.json]
"""
import csv, glob, json, os, re, sys


def short(k):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", k)
    return re.sub(r"\(.*$", "", k)


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    marks = [i for i, e in enumerate(ev) if "advance_iter" in e[0]]
    segs = [(marks[i] + 1, marks[i + 1] + 1) for i in range(len(marks) - 1)]
    if not segs:
        print("no step markers")
        return
    # the steady (graph-replayed) steps all have the same launch count: the MOST COMMON one (the very last segment may carry tear-down copies)
    import collections
    n_last = collections.Counter(s[1] - s[0] for s in segs).most_common(1)[0][0]
    if os.environ.get("TRACE_LAST_GROUP") and len(segs) > 2:       # a run with two workloads (bench.py --shard-emulate: the shard's steps come last)
        n_last = segs[-2][1] - segs[-2][0]
    steady = [s for s in segs if s[1] - s[0] == n_last][-20:]
    if os.environ.get("TRACE_LAST_STEPS"):          # eager workloads (tools/bench_train.py): the launch count may differ by step
        steady = segs[-int(os.environ["TRACE_LAST_STEPS"]):]
    by, busy, span = {}, 0.0, 0.0
    for a, b in steady:
        seg = ev[a:b]
        span += seg[-1][2] - ev[a - 1][2]          # from the end of the previous step's last kernel to the end of this one
        for k, s, e in seg:
            x = by.setdefault(k, [0, 0.0])
            x[0] += 1
            x[1] += e - s
            busy += e - s
    n = len(steady)
    print(f"{n} steady steps of {n_last} launches: span {span / n / 1e3:.1f} us/step, kernel-busy {busy / n / 1e3:.1f} us/step, "
          f"idle between kernels {(span - busy) / n / 1e3:.1f} us/step ({(span - busy) / n / n_last:.0f} ns per launch)")
    out = {"steps": n, "launches_per_step": n_last, "span_us_per_step": span / n / 1e3, "busy_us_per_step": busy / n / 1e3, "kernels": {}}
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k[:110]:110s} x{c / n:6.1f}  {t / n / 1e3:9.1f} us/step  avg {t / c / 1e3:7.2f} us  {100 * t / busy:5.1f}%")
        out["kernels"][k] = {"launches_per_step": c / n, "us_per_step": t / n / 1e3, "avg_us": t / c / 1e3}
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
