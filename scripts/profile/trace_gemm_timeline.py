"""Per-workgroup timeline of the ring-GEMM launches of ONE headline iteration (development aid).

Needs a trace build of the library, built OUTSIDE the tree on the GPU box (ablation / trace libraries are never shipped):
    make -C pxmcmc_amd/csrc BUILD=/tmp/pxm_ab/build_tr OUT=/tmp/pxm_ab/libpxm_tr.so EXTRA="-DPXM_GEMM_TRACE -DPXM_D5_TRACE"
    PXM_LIB_PATH=/tmp/pxm_ab/libpxm_tr.so python scripts/profile/trace_gemm_timeline.py     (TRACE=dft: the grouped DFT launch)
Every workgroup of k_sht_gemm records (block id, grid, start, end [100 MHz wall clock], XCC / SE / CU, chunks, row tiles,
Gram flag).  Printed per launch: span, per-workgroup duration against its chunk count, workgroups per CU, idle share.
"""
import collections, contextlib, ctypes as C, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pxmcmc_amd import ops
from pxmcmc_amd._lib import lib
from pxmcmc_amd.forward import SphericalWaveletTransformOperator
from pxmcmc_amd.mcmc import MYULA, PxMCMCParams
from pxmcmc_amd.prior import S2_Wavelets_L1

L, B, J_MIN, Cn = bench.L, bench.B, bench.J_MIN, int(os.environ.get("C", bench.CHAINS_PER_GPU))
sht = ops.ShtPlan(L, 0, max_chains=1)
truth, rng = bench.synthetic_field(lambda flm: sht.inverse(flm).cpu().numpy(), L, seed=2)
del sht
data = truth + bench.SIGMA * rng.normal(size=truth.size)
op = SphericalWaveletTransformOperator(data, bench.SIGMA, "synthesis", L, B, J_MIN, max_chains=Cn)
reg = S2_Wavelets_L1("synthesis", op.transform.inverse, op.transform.inverse_adjoint, bench.LMDA * bench.MU, L=L, B=B, J_min=J_MIN)
delta, _ = bench.stable_delta(op.transform, bench.SIGMA, bench.LMDA)
params = PxMCMCParams(lmda=bench.LMDA, delta=delta, mu=bench.MU, nsamples=1, nburn=0, ngap=1, verbosity=0)
s = MYULA(op, reg, params, nchains=Cn, rng="philox", seed=2, use_graph=False)
s._prepare()
with contextlib.redirect_stdout(io.StringIO()):
    X, preds = s._initial_sample(np.zeros(op.nparams))
if s._pairs_ok(X):
    s._pairs_start()
s._engine_start(X, preds, 0)
s._engine_advance(40)
torch.cuda.synchronize()

NREC = 8192  # (DFT trace: workgroup records in the first 4096 slots, phase records behind them; GEMM: chunk stamps behind 8192)
buf = torch.zeros(8 + 8 * 2 * NREC, dtype=torch.int64, device="cuda")
if os.environ.get("TRACE") == "dft":  # grouped DFT kernel instead (-DPXM_D5_TRACE build)
    fn = lib.pxm_debug_set_dft_trace
    fn.argtypes = [C.c_void_p]
    assert fn(C.c_void_p(buf.data_ptr())) == 0
    s._engine_advance(2)
    torch.cuda.synchronize()
    buf.zero_()
    s._engine_advance(1)
    torch.cuda.synchronize()
    h = buf.cpu().numpy().astype(np.uint64)
    n = int(h[0])
    a = h[8:8 + 8 * n].reshape(n, 8).astype(np.int64)
    t0, t1 = a[:, 2].min(), a[:, 3].max()
    dur, start, ends = (a[:, 3] - a[:, 2]) / 100.0, (a[:, 2] - t0) / 100.0, (a[:, 3] - t0) / 100.0
    hw, xcc = a[:, 4] & 0xffffffff, a[:, 4] >> 32
    cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
    print(f"k_ring2px_group5: {n} workgroups recorded (grid {int(a[0, 1])}), span {(t1 - t0) / 100.0:.1f} us, CUs used {len(set(cu.tolist()))}")
    print("  scale entry (r0): workgroups, duration us min / median / max, start median / max, end median / max")
    for e in sorted(set(a[:, 6].tolist())):
        sel = a[:, 6] == e
        print(f"  entry {e} (r0={int(a[sel, 5][0])}): {int(sel.sum()):5d}  {dur[sel].min():5.1f} / {np.median(dur[sel]):5.1f} / {dur[sel].max():5.1f}"
              f"   start {np.median(start[sel]):5.1f} / {start[sel].max():5.1f}   end {np.median(ends[sel]):5.1f} / {ends[sel].max():5.1f}")
    busy = collections.defaultdict(float)
    for c_, d_ in zip(cu.tolist(), dur.tolist()):
        busy[c_] += d_
    b = np.array(list(busy.values()))
    print(f"  workgroup-time per CU: min {b.min():.0f} / median {np.median(b):.0f} / max {b.max():.0f} us (2 workgroups resident per CU: "
          f"ideal span = total / (2 x 256) = {dur.sum() / 512:.1f} us)")
    for q in (50, 75, 90, 95, 99, 100):
        print(f"  {q:3d} % of the workgroups have finished by {np.percentile(ends, q):5.1f} us")
    m = int(h[1])
    ph = h[8 + 8 * 4096:8 + 8 * 4096 + 8 * m].reshape(m, 8).astype(np.int64)
    print(f"  phases (us from workgroup start; median over {m} workgroups): r0: rings staged / inverse transform / update / forward transform / rings stored")
    for r0 in sorted(set(ph[:, 0].tolist())):
        q = ph[ph[:, 0] == r0]
        if r0 == 9:  # exact-length body (csrc/dft_pfa.h): its own stamps
            med = np.median(q[:, 1:8], axis=0) / 100.0
            print("    r0=9 (exact length): rings staged / unit gathered / inverse transform / first half of the epilogue / second half / "
                  "forward transform / rings stored:")
            print("          " + " / ".join(f"{v:5.1f}" for v in med) + f"   ({len(q)} workgroup passes)")
            continue
        med = np.median(q[:, 1:6], axis=0) / 100.0
        print(f"    r0={r0}: " + " / ".join(f"{v:5.1f}" for v in med) + f"   ({len(q)} workgroups)")
    sys.exit(0)
fn = lib.pxm_debug_set_gemm_trace  # (only in a -DPXM_GEMM_TRACE build)
fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(buf.data_ptr())) == 0
s._engine_advance(2)
torch.cuda.synchronize()
buf.zero_()
s._engine_advance(2)  # two iterations = six launches
torch.cuda.synchronize()
h = buf.cpu().numpy().astype(np.uint64)
n = int(h[0])
rec = h[8:8 + 8 * n].reshape(n, 8).astype(np.int64)
print(f"{n} workgroup records")
# split into launches by (grid, gram flag) runs in start-time order
order = np.argsort(rec[:, 2])
rec = rec[order]
launches = []
for r in rec:
    key = (int(r[1]), int(r[5] >> 32))
    if launches and launches[-1][0] == key and len(launches[-1][1]) < key[0]:
        launches[-1][1].append(r)
    else:
        launches.append((key, [r]))
for (grid, gram), rows in launches:
    a = np.array(rows)
    nchs = a[:, 5] & 0xffff
    t_stage, t_loop = (a[:, 6] - a[:, 2]) / 100.0, (a[:, 7] - a[:, 2]) / 100.0
    t0, t1 = a[:, 2].min(), a[:, 3].max()
    dur = (a[:, 3] - a[:, 2]) / 100.0  # us
    start = (a[:, 2] - t0) / 100.0
    hw, xcc = a[:, 4] & 0xffffffff, a[:, 4] >> 32
    cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
    per_cu = collections.Counter(cu.tolist())
    busy = collections.defaultdict(float)
    for c, d in zip(cu.tolist(), dur.tolist()):
        busy[c] += d
    span = (t1 - t0) / 100.0
    print(f"\nlaunch grid={grid} gram={gram}: {len(a)} records, span {span:.1f} us; CUs used {len(per_cu)}, workgroups per CU "
          f"{dict(collections.Counter(per_cu.values()))}; start offsets: median {np.median(start):.1f} us, max {start.max():.1f} us")
    print("  chunks : workgroups, duration us (min / median / max), us per chunk (median)")
    for nch in sorted(set(nchs.tolist())):
        sel = nchs == nch
        d = dur[sel]
        if sel.sum() >= 1 and (nch in (1, 2, 4, 8, 12, 16) or nch == nchs.max()):
            print(f"  {nch:6d} : {int(sel.sum()):5d}  {d.min():6.1f} / {np.median(d):6.1f} / {d.max():6.1f}   {np.median(d) / max(nch, 1):5.2f}"
                  f"   first chunk staged at {np.median(t_stage[sel]):5.1f} us, loop done at {np.median(t_loop[sel]):5.1f} us, epilogue {np.median(d - t_loop[sel]):5.1f} us")
    ends = (a[:, 3] - t0) / 100.0
    print(f"  last workgroup to finish: block {int(a[np.argmax(ends), 0])} with {int(nchs[np.argmax(ends)])} chunks, started at {start[np.argmax(ends)]:.1f} us, "
          f"ran {dur[np.argmax(ends)]:.1f} us; 90 % of the workgroups are done by {np.percentile(ends, 90):.1f} us")

m = int(h[1])
q = h[8 + 8 * 8192:8 + 8 * 8192 + 8 * m].reshape(m, 8).astype(np.int64)
print("\nchunk 8 of the tasks that have one (thread 0 of the workgroup; shader-clock cycles, median): operand wait + LDS store / barrier / B reads + MFMA issue / loop tail")
for key in sorted(set(zip(q[:, 6].tolist(), q[:, 5].tolist(), q[:, 0].tolist()))):
    sel = (q[:, 6] == key[0]) & (q[:, 5] == key[1]) & (q[:, 0] == key[2])
    med = np.median(q[sel, 1:5], axis=0)
    print(f"  grid {key[0]} gram={key[1]} chunks={key[2]}: " + " / ".join(f"{int(v):6d}" for v in med) + f"   sum {int(med.sum())} cycles ({int(sel.sum())} workgroups)")
