"""numpy check of the Riccati form of the condensed Hessian's factor (design study for lmpc_fused_ric.hpp):
   H^-1 = Ginv' ... with R^-1 = Gamma blkdiag(Lam_k^-T):  the forward / adjoint closed-loop recursions reproduce
   z = R^-1 v and w = R^-T n, R'R = H, and the LQR roll-out reproduces -H^-1 c."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle
from copra_amd import workloads

wl = workloads.com_preview(4)
k0 = 1
A, B, d, x0, N = wl["A"][k0], wl["B"][k0], wl["d"][k0], wl["x0"][k0], wl["N"]
nx, nu = 6, 3
qp = pyoracle.lmpc_build(A, B, d, x0, N, wl["costs"], wl["cstrs"])
H, c = qp["Q"], qp["c"]
# stage data: W_in (k < N), W_N, q_in, q_N
wx = np.array(wl["costs"][0]["weights"]); px = np.array(wl["costs"][0]["p"])
wu = np.array(wl["costs"][1]["weights"]); pu = np.array(wl["costs"][1]["p"])
Win = np.zeros((9, 9)); Win[:6, :6] = np.diag(wx); Win[6:, 6:] = np.diag(wu) + 1e-6 * np.eye(3)
qin = np.concatenate([-wx * px, -wu * pu])
WN = np.diag(wx); qN = -wx * px
AB = np.hstack([A, B])
P, p = WN.copy(), qN.copy()
Acl, Bt, K, Li, kv = [None] * N, [None] * N, [None] * N, [None] * N, [None] * N
for k in range(N - 1, -1, -1):
    M = Win + AB.T @ P @ AB
    h = qin + AB.T @ (P @ d + p)
    Huu, Hux, Hxx = M[6:, 6:], M[6:, :6], M[:6, :6]
    Lam = np.linalg.cholesky(Huu)
    Li[k] = np.linalg.inv(Lam)
    K[k] = -np.linalg.solve(Huu, Hux)
    kv[k] = -np.linalg.solve(Huu, h[6:])
    Acl[k] = A + B @ K[k]
    Bt[k] = B @ Li[k].T
    P = Hxx + Hux.T @ K[k]
    p = h[:6] + Hux.T @ kv[k]
# unconstrained minimiser by roll-out
x = x0.copy(); U = []
for k in range(N):
    u = K[k] @ x + kv[k]; U.append(u); x = A @ x + B @ u + d
U = np.concatenate(U)
print("x_unc vs -H^-1 c:", np.abs(U + np.linalg.solve(H, c)).max())
def op_z(v):  # z = R^-1 v = Gamma blkdiag(Lam^-T) v
    xi = np.zeros(nx); out = []
    for k in range(N):
        vk = v[3 * k:3 * k + 3]
        out.append(K[k] @ xi + Li[k].T @ vk)
        xi = Acl[k] @ xi + Bt[k] @ vk
    return np.concatenate(out)
def op_w(nv):  # w = R^-T n = blkdiag(Lam^-1) Gamma' n
    mu = np.zeros(nx); out = [None] * N
    for k in range(N - 1, -1, -1):
        nk = nv[3 * k:3 * k + 3]
        out[k] = Li[k] @ nk + Bt[k].T @ mu
        mu = Acl[k].T @ mu + K[k].T @ nk
    return np.concatenate(out)
rng = np.random.default_rng(0)
v = rng.standard_normal(60); nv = rng.standard_normal(60)
Rinv = np.column_stack([op_z(e) for e in np.eye(60)])
print("R^-1 R^-T = H^-1:", np.abs(Rinv @ Rinv.T - np.linalg.inv(H)).max() / np.abs(np.linalg.inv(H)).max())
print("op_w = R^-T n:", np.abs(op_w(nv) - Rinv.T @ nv).max())
