"""accuracy of the device fp64 Box-Muller against an 80-bit long-double evaluation (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pxmcmc_amd import ops

rng = np.random.default_rng(0)
n = 1 << 20
u1 = np.concatenate([rng.random(n), 1 - rng.random(1000) * 1e-12, rng.random(1000) * 1e-12 + 1e-17, [1.0, 0.5 * 2.0 ** -53, 1 - 2.0 ** -53]])
u2 = np.concatenate([rng.random(n), rng.random(2000), [0.3, 0.3, 0.3]])
z0, z1 = [z.cpu().numpy() for z in ops.box_muller(u1, u2, noise64=True)]
L = np.longdouble
rad = np.sqrt(-2 * np.log(u1.astype(L)))
t = 4 * u2.astype(L)
k = np.rint(t)
a = (t - k) * (np.pi * L(1) / 2 + L(6.123233995736766036e-17) / 1)  # pi/2 to ~1e-33 in two pieces
a = (t - k) * (L(1.5707963267948966192313216916397514))
cs, sn = np.cos(a), np.sin(a)
q = k.astype(int) % 4
C = np.where(q == 0, cs, np.where(q == 1, -sn, np.where(q == 2, -cs, sn)))
S = np.where(q == 0, sn, np.where(q == 1, cs, np.where(q == 2, -sn, -cs)))
w0, w1 = (rad * C), (rad * S)
e0 = np.abs(z0 - w0).astype(float); e1 = np.abs(z1 - w1).astype(float)
scale = np.maximum(rad.astype(float), 1e-300)
print("max abs err", e0.max(), e1.max())
print("max err / radius", (e0 / scale).max(), (e1 / scale).max(), "at", int((e0 / scale).argmax()), u1[(e0 / scale).argmax()], u2[(e0 / scale).argmax()])
r_dev = np.hypot(z0.astype(L), z1.astype(L))
print("radius rel err max", float((np.abs(r_dev - rad)[rad > 0] / rad[rad > 0]).max()))
print("finite", np.isfinite(z0).all() and np.isfinite(z1).all(), "u1=1:", z0[-3], z1[-3])
