#!/bin/bash
# Per-DISPATCH view of one graph-replayed step: rocprofv3 --kernel-trace of a short bench run, the dispatches between the last two
# cfg_ddim_kernel launches with their durations and full grid (tiles x split-K for the GEMMs) -> gpurun_out/<tag>/dispatches.csv and a
# summary grouped by (kernel, grid).  Finds the launches whose grid under-fills the chip (how the low-resolution GroupNorm-apply was found).
tag=${1:-dispatch}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktr
rocprofv3 --kernel-trace -d /tmp/ktr -o t --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary "${@:2}" > /tmp/ktr.log 2>&1
python - "$R/gpurun_out/$tag" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob('/tmp/ktr/**/*kernel_trace.csv', recursive=True)[0])))
idx = [i for i, r in enumerate(rows) if 'cfg_ddim' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
agg = collections.OrderedDict()
with open(out + '/dispatches.csv', 'w') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dur_us', 'wgs_x', 'wgs_y', 'wgs_z', 'wg_size', 'lds', 'vgpr'])
    for r in step:
        name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
        g = [int(r['Grid_Size_' + a]) // int(r['Workgroup_Size_' + a]) for a in 'XYZ']
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        w.writerow([name, f'{dur:.2f}'] + g + [int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']),
                                              r['LDS_Block_Size'], r['VGPR_Count']])
        a = agg.setdefault((name, tuple(g)), [0, 0.0])
        a[0] += 1
        a[1] += dur
with open(out + '/dispatch_summary.txt', 'w') as f:
    f.write(f'{len(step)} dispatches, kernel-busy {sum(a[1] for a in agg.values()):.1f} us\n')
    for (name, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f'{name:46s} grid {g[0]:5d} x {g[1]:3d} x {g[2]:2d} = {g[0] * g[1] * g[2]:6d} wgs  x{c:3d}  total {t:8.1f} us  avg {t / c:7.2f}\n')
print(open(out + '/dispatch_summary.txt').read()[:3000])
PY
