import os, sys, faulthandler, numpy as np
faulthandler.dump_traceback_later(45, exit=True)
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault("LGH_VCG_VARIANT", "4")
seed = int(sys.argv[1]); tol = float(sys.argv[2]); timers = int(sys.argv[3])
from helpers import make_gpu, make_oracle, deformed_state, rel_err
from oracle.fem import Problem
prob = Problem(mesh="box01_hex", rs=1, order_v=3, order_e=2, problem=1)
H1V = prob.H1V; N = H1V // 3
S = deformed_state(prob, seed=seed)
o = make_oracle(prob); o.cg_tol = tol
dS_o = np.empty_like(S); o.qdata_is_current = False; o.reset_timers(); o.mult(S, dS_o); ito = o.timers()["H1iter"]
print("oracle done", ito, flush=True)
g = make_gpu(prob); g.ctx.enable_timers(bool(timers)); g.cg_tol = tol
Sd = g.ctx.to_dev(S); dS = g.ctx.zeros(S.size); g.reset_quadrature_data(); g.ctx.reset_timers()
print("gpu set up", flush=True)
g.mult(Sd, dS); print("mult enqueued", flush=True); g.ctx.sync(); print("synced", flush=True)
itg = g.ctx.timers()["H1iter"]; got = dS.cpu().numpy(); rhs = g.rhs.cpu().numpy()
for k in range(3):
    x_o, it_o = o.cg(0, rhs[k * N:(k + 1) * N].copy(), comp=k, rel_tol=tol, max_iter=300)
    print("comp", k, "oracle cg on the gpu's rhs: iters", it_o, "| gpu vs that %.2e" % rel_err(got[H1V + k * N:H1V + (k + 1) * N], x_o),
          "| oracle mult vs that %.2e" % rel_err(dS_o[H1V + k * N:H1V + (k + 1) * N], x_o), flush=True)
diag = np.array(o.diagV)
for k in range(3):
    b = rhs[k * N:(k + 1) * N].copy()
    ess = np.asarray(prob.ess[k], dtype=np.int64)
    dg = diag.copy()
    if len(ess): dg[ess] = 1.0
    A = lambda v: o.mass_mult(0, v, comp=k)
    x = np.zeros(N); r = b.copy(); z = r / dg; d = z.copy(); nom = float(d @ r); r0 = nom * tol * tol
    xs = {}
    gk, ok_ = got[H1V + k * N:H1V + (k + 1) * N], dS_o[H1V + k * N:H1V + (k + 1) * N]
    for i in range(1, 40):
        Ad = A(d); den = float(d @ Ad); al = nom / den
        x = x + al * d; r = r - al * Ad; z = r / dg; bn = float(r @ z)
        xs[i] = x.copy()
        if i >= 20:
            print("  comp", k, "it", i, "rz/r0 %.4f" % (bn / r0), "| gpu vs x_i %.2e" % rel_err(gk, x), "| oracle mult vs x_i %.2e" % rel_err(ok_, x), flush=True)
        if bn <= r0 * 1e-4: break
        d = z + (bn / nom) * d; nom = bn
print("seed", seed, "tol", tol, "timers", timers, "iters oracle/gpu", ito, itg, flush=True)
g.close(); o.close()
