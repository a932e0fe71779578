"""Summarise tools/pmc_mfma.sh: per kernel (steady-state launches only), MFMA-pipe utilisation and the wave-time split.

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip (16 cycles per 16x16x32 MFMA); GRBM_GUI_ACTIVE is summed
over the 8 XCDs.  util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).  SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY
(issue stalls) and SQ_ACTIVE_INST_ANY partition SQ_WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import csv, glob, json, os, re, sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_traffic import steady_rows
    rows = steady_rows(rows)                            # graph-replayed steps only (no autotuner launches)
    acc = {}
    for r in rows:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k)
        a = acc.setdefault(k, {})
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    res = {}
    for k, a in acc.items():
        g = a.get("GRBM_GUI_ACTIVE", [1, 0.0])
        n = g[0]
        gui = g[1] / 8.0
        mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", [1, 0.0])[1]
        wc = a.get("SQ_WAVE_CYCLES", [1, 1.0])[1] or 1.0
        res[k] = {"launches_profiled": n, "gui_active_cycles_per_launch": gui / n,
                  "mfma_pipe_util": mf / (gui * 1024.0) if gui else 0.0,
                  "wave_time_parked": a.get("SQ_WAIT_ANY", [1, 0.0])[1] / wc,
                  "wave_time_issue_stall": a.get("SQ_WAIT_INST_ANY", [1, 0.0])[1] / wc,
                  "wave_time_issuing": a.get("SQ_ACTIVE_INST_ANY", [1, 0.0])[1] / wc}
    json.dump({"recipe": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                         "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 12 --warmup 1 --no-cpu-baseline; last third of the dispatches",
               "kernels": res}, open(out, "w"), indent=1)
    tot = sum(v["gui_active_cycles_per_launch"] * v["launches_profiled"] for v in res.values())
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_launch"] * kv[1]["launches_profiled"])[:14]:
        share = v["gui_active_cycles_per_launch"] * v["launches_profiled"] / tot
        print(f"{k[:50]:50s} n={v['launches_profiled']:5d} time {100 * share:4.1f}%  MFMA util {100 * v['mfma_pipe_util']:5.1f}%  parked "
              f"{100 * v['wave_time_parked']:4.1f}% stall {100 * v['wave_time_issue_stall']:4.1f}% issuing {100 * v['wave_time_issuing']:4.1f}%")


if __name__ == "__main__":
    main()
