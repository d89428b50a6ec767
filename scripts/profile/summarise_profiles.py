#!/usr/bin/env python
"""Turn raw rocprofv3 output (gpurun_out/<tag>/...) into the committed summaries under profiles/.

  python scripts/profile/summarise_profiles.py r01a r01
writes profiles/<name>_kernel_stats.csv (rocprofv3 --stats table, verbatim),
       profiles/<name>_summary.md       (per-kernel table + GEMM launch classes + HBM traffic),
       profiles/pmc_summary.json         (HBM bytes per k_sht_gemm launch, read by bench.py).
HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so the
read side is doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
sq_tag = sys.argv[3] if len(sys.argv) > 3 else None   # gpurun_out/<sq_tag>/{a,b,c}: SQ counter passes (collect.sh sq)
c5_tag = sys.argv[4] if len(sys.argv) > 4 else None   # gpurun_out/<c5_tag>: kernel trace of scripts/timing/time_config5.py (collect.sh config5)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)))
    return hits[-1] if hits else None


stats = one("trace/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{name}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
trace = list(csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))))
bench_line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")]
bench = json.loads(bench_line[-1]) if bench_line else {}

classes = collections.defaultdict(list)
for r in trace:
    if "k_sht_gemm" in r["Kernel_Name"]:
        classes[int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def pmc(counter, sub):
    f = one(f"{sub}/*/*_counter_collection.csv")
    per = collections.defaultdict(list)
    if not f:
        return per
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


fetch, write = pmc("FETCH_SIZE", "pmc_fetch"), pmc("WRITE_SIZE", "pmc_write")
out = [f"# rocprofv3 summary `{name}` (raw: gpurun_out/{tag}, command: bench.py --steps 100 --warmup 10)\n"]
if bench:
    out.append(f"bench line under the profiler: {bench['value']:.0f} samples/s, {bench['ms_per_step']:.3f} ms/step, "
               f"GEMM avg launch {bench['roofline']['avg_launch_us']:.1f} us (live HIP events)\n")
out.append("## kernel-trace --stats (top kernels)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:14]:
    out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
out.append("\n## k_sht_gemm launch classes (by workgroup count)\n\n| workgroups | launches | avg us |\n|---|---|---|")
for k, v in sorted(classes.items()):
    if len(v) >= 10:
        out.append(f"| {k} | {len(v)} | {sum(v)/len(v):.1f} |")
summary = {}
# the k_sht_gemm variant of the timed steps = the one with the largest total time in the kernel trace
gemm_rows = [r for r in rows if "k_sht_gemm<" in r["Name"]]
dominant = max(gemm_rows, key=lambda r: float(r["TotalDurationNs"]))["Name"].split("(")[0].strip() if gemm_rows else ""
out.append("\n## HBM traffic per launch from PMC (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE as is; KiB -> bytes)\n\n| kernel | launches | read MB | write MB | total MB |\n|---|---|---|---|---|")
for k in sorted(fetch):
    if k in write and len(fetch[k]) >= 4:
        rd = 2 * 1024 * sum(fetch[k]) / len(fetch[k])
        wr = 1024 * sum(write[k]) / len(write[k])
        out.append(f"| `{k[:60]}` | {len(fetch[k])} | {rd/1e6:.1f} | {wr/1e6:.1f} | {(rd+wr)/1e6:.1f} |")
        # the variants the timed steps launch: same <CT, NSLAB, NW, RT, NSET> as the dominant one, any operand flags
        # (Gram: second operand, forward-adjoint: row scale, forward: neither) -- averaged over all their launches
        if "k_sht_gemm<" in k and k.strip().split(",")[:5] == dominant.split(",")[:5]:
            n0 = summary.get("launches_sampled", 0)
            n1 = len(fetch[k])
            for key, val in (("k_sht_gemm_hbm_bytes_per_launch", rd + wr), ("k_sht_gemm_read_bytes", rd), ("k_sht_gemm_write_bytes", wr)):
                summary[key] = (summary.get(key, 0.0) * n0 + val * n1) / (n0 + n1)
            summary["kernel"] = ",".join(dominant.split(",")[:5]) + ", *, *> (all operand-flag variants of the timed steps)"
            summary["launches_sampled"] = n0 + n1
if bench.get("roofline", {}).get("launch_classes"):
    out.append("\n## k_sht_gemm launch classes from the bench line (live HIP events, algorithmic bytes per launch)\n\n| alg MB | launches | avg us | TB/s | of 8 TB/s |\n|---|---|---|---|---|")
    for c in bench["roofline"]["launch_classes"]:
        out.append(f"| {c['alg_MB']:.1f} | {c['launches']} | {c['avg_us']:.1f} | {c['GBs']/1e3:.2f} | {c['frac']:.2f} |")
if sq_tag:
    out.append("\n## SQ counters per launch (rocprofv3 --pmc, three passes; sums over the chip)\n")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in "abc":
        f = sorted(glob.glob(os.path.join(root, "gpurun_out", sq_tag, sub, "*", "*_counter_collection.csv")))
        if not f:
            continue
        for r in csv.DictReader(open(f[-1])):
            k = r["Kernel_Name"].split("(")[0]
            if "k_ring2px_group" in k:
                acc[k.strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
            elif "k_sht_gemm<1, 2" in k:
                acc[k.strip() + f" grid {r['Grid_Size']}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        out.append(f"* `{k}`: " + ", ".join(f"{c} {v:,.0f}" for c, v in sorted(m.items())))
        if "SQ_BUSY_CYCLES" in m and "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
            cyc = m["SQ_BUSY_CYCLES"] / 32  # summed over 8 XCDs x 4 shader engines
            out.append(f"  - kernel = {cyc:,.0f} cycles; VALU issue = {m['SQ_INSTS_VALU'] * 4 / 1024:,.0f} cycles per SIMD = "
                       f"{m['SQ_INSTS_VALU'] * 4 / 1024 / cyc:.2f} of the kernel; {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:,.0f} VALU instructions per wave"
                       + (f"; waves wait (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f} of their cycles"
                          if "SQ_WAVE_CYCLES" in m else "")
                       + (f"; LDS busy {m['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.2f} of the kernel per CU, bank-conflict cycles "
                          f"{m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f} of them" if "SQ_LDS_IDX_ACTIVE" in m else ""))
if c5_tag:
    f = sorted(glob.glob(os.path.join(root, "gpurun_out", c5_tag, "*", "*_kernel_stats.csv")))
    if f:
        shutil.copy(f[-1], os.path.join(dst, f"{name}_L512_kernel_stats.csv"))
        rows5 = list(csv.DictReader(open(f[-1])))
        log = [l.strip() for l in open(os.path.join(root, "gpurun_out", c5_tag + ".log")) if l.startswith("C=")]
        o5 = [f"# rocprofv3 --kernel-trace --stats of scripts/timing/time_config5.py (BASELINE configs[4] sizes: L=512 weak lensing, PxMALA, eager launches)\n",
              *[f"    {l}" for l in log], "\n| kernel | calls | avg us | % |\n|---|---|---|---|"]
        for r in rows5[:24]:
            o5.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
        open(os.path.join(dst, f"{name}_L512_summary.md"), "w").write("\n".join(o5) + "\n")
open(os.path.join(dst, f"{name}_summary.md"), "w").write("\n".join(out) + "\n")
if summary:
    summary["source"] = f"profiles/{name}_summary.md"
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print("\n".join(out))
