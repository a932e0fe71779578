"""Summarise a rocprofv3 --pmc pass over tools/gemm_probe.py (MVD_PROBE_PLAIN=1): per (kernel, grid) average counters."""
import csv, glob, os, re, sys


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    acc = {}
    for r in rows:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k)
        if "gemm_kernel" not in k and "splitk" not in k:
            continue
        key = (k, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        a = acc.setdefault(key, {})
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    for key, a in acc.items():
        n = max(c[0] for c in a.values())
        vals = {name: c[1] / c[0] for name, c in a.items()}
        line = f"{key[0][:44]:44s} grid {key[1]:>8s} n={n:3d} "
        if "GRBM_GUI_ACTIVE" in vals:
            gui = vals["GRBM_GUI_ACTIVE"] / 8.0
            line += f"gui {gui:9.0f} cyc "
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
                line += f"mfma_util {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024.0) * 100:5.1f}% "
        if "SQ_WAVE_CYCLES" in vals:
            wc = vals["SQ_WAVE_CYCLES"]
            for nm, lab in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "stall"), ("SQ_ACTIVE_INST_ANY", "issuing")):
                if nm in vals:
                    line += f"{lab} {vals[nm] / wc * 100:4.1f}% "
        for nm in sorted(vals):
            if nm not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                line += f"{nm}={vals[nm]:.3g} "
        print(line)


if __name__ == "__main__":
    main()
