"""Per-launch durations of ONE kernel inside the graph-replayed steps of a rocprofv3 --kernel-trace, grouped by launch geometry.
    python tools/kernel_shapes.py <trace dir> <kernel name substring>"""
import csv, glob, os, sys, collections


def main():
    d, name = sys.argv[1], sys.argv[2]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "advance_iter" in r["Kernel_Name"]]
    lo = marks[-21] if len(marks) > 21 else marks[0]
    sel = [r for r in rows[lo:marks[-1]] if name in r["Kernel_Name"]]
    steps = max(1, min(20, len(marks) - 1))
    by = collections.defaultdict(list)
    for r in sel:
        key = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
        by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in by.values())
    print(f"{name}: {len(sel) / steps:.1f} launches / step, {tot / steps:.1f} us / step")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"  grid {k[0]:>8s} x {k[1]:>5s}  LDS {k[2]:>7s}: x{len(v) / steps:5.1f} / step  avg {sum(v) / len(v):7.2f} us  min {min(v):7.2f}  total {sum(v) / steps:8.1f} us / step")


if __name__ == "__main__":
    main()
