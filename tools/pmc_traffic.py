"""Summarise the two rocprofv3 --pmc passes of tools/pmc_traffic.sh: average FETCH_SIZE / WRITE_SIZE per launch and kernel.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads (global_load_dwordx4 and LDS-DMA alike) at 64 bytes (MI355X_MICROARCH.md, HBM).
"""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def steady_rows(rows):
    """Dispatch rows of the graph-REPLAYED steps only: steps end with advance_iter_kernel; keep the trailing run of steps that
    have the same number of dispatches as the last one (drops weight packing, the eager warm-up step and every autotuner launch)."""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    name = {}
    for r in rows:
        name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    marks = [i for i in ids if "advance_iter" in name[i]]
    if len(marks) < 3:
        return rows[len(rows) * 2 // 3:]
    segs = [(marks[i] + 1, marks[i + 1]) for i in range(len(marks) - 1)]
    n_last = sum(1 for i in ids if segs[-1][0] <= i <= segs[-1][1])
    keep = set()
    for a, b in reversed(segs):
        seg = [i for i in ids if a <= i <= b]
        if len(seg) != n_last:
            break
        keep.update(seg)
    return [r for r in rows if int(r["Dispatch_Id"]) in keep]


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, f"no counter_collection.csv under {d}"
    acc = {}
    rows = steady_rows(list(csv.DictReader(open(f[0]))))
    for row in rows:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", row["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for k in fetch:
        nf, sf = fetch[k]
        nw, sw = write.get(k, [1, 0.0])
        res[k] = {"launches_profiled": nf, "fetch_size_kib_per_launch": sf / nf, "write_size_kib_per_launch": sw / max(nw, 1),
                  "hbm_bytes_per_launch": (2.0 * sf / nf + sw / max(nw, 1)) * 1024.0}
    import bench
    json.dump({"tree": bench.tree_fingerprint(), "recipe": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 12 --warmup 1 "
                         "--no-cpu-baseline [workload flags]; graph-replayed steps only; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
               "workload_args": sys.argv[4:], "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"])[:12]:
        print(f"{k[:60]:60s} n={v['launches_profiled']:5d}  {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
