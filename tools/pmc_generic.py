"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass of bench.py collected (steady-state launches only).
    python tools/pmc_generic.py <dir with *counter_collection.csv> <out.json>"""
import csv, glob, json, os, re, sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_traffic import steady_rows
    rows = steady_rows(rows)
    acc = {}
    for r in rows:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k)
        c = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    res = {k: {"launches_profiled": max(v[0] for v in a.values()), **{n: v[1] / v[0] for n, v in a.items()}} for k, a in acc.items()}
    json.dump({"kernels": res}, open(out, "w"), indent=1)
    names = sorted({n for a in acc.values() for n in a})
    print("kernel".ljust(46) + " ".join(n[-22:].rjust(22) for n in names))
    key = names[0]
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get(key, 0) * kv[1]["launches_profiled"])[:16]:
        print(k[:45].ljust(46) + " ".join(f"{v.get(n, 0):22.4g}" for n in names))


if __name__ == "__main__":
    main()
