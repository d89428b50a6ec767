"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: start / end of every kernel of the last full
iterations relative to the iteration's first kernel (which lanes overlap, where the gaps are).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 40 --warmup 5 --ramp 0 ...
    python scripts/profile/step_timeline.py OUT [n_steps]
"""
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void pxm::", "").replace("pxm::", "")
    return name.split("(")[0][:46]


def main():
    out = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = sorted(glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), grid // max(wg, 1),
                         r.get("Queue_Id", "?")))
    rows.sort()
    # an iteration starts at the Gram launch (the two-operand GEMM variant with the affine epilogue: "true, false>")
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_sht_gemm") and "true, false>" in r[2]]
    if len(starts) < nsteps + 2:
        print("not enough iterations in the trace")
        return
    # the middle of the run: replayed graph iterations
    mid = len(starts) // 2
    for k in range(nsteps):
        a, b = starts[mid + k], starts[mid + k + 1]
        t0 = rows[a][0]
        print(f"--- iteration {mid + k}: {(rows[b][0] - t0) / 1e3:7.1f} us to the next Gram launch")
        for s, e, name, wgs, q in rows[a:b]:
            print(f"  {(s - t0) / 1e3:7.1f} -> {(e - t0) / 1e3:7.1f} us  ({(e - s) / 1e3:6.1f})  q{q:>3}  wgs {wgs:5d}  {name}")


if __name__ == "__main__":
    main()
