"""Debug aid: compares the number of columns the bound pricing lists on the GPU with a numpy model of the same rule."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lp = importlib.import_module("linear-programming_amd")
f32 = np.float32
m, n, upto = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = n + m
eng = lp.TableauEngine(m, N)
eng.fill_synthetic(n, 1)
prev = (0, 0)
for it in range(upto):
    T, obj, basic = eng.download()
    rhs = T[:, N].copy()
    cand = (np.arange(N + 2) != N) & (basic == 0) & (obj.astype(f32) >= f32(1e-6)); cand[N + 1] = False
    cj = np.nonzero(cand)[0]
    nblk = min((m + 255) // 256, 256)
    blkmin = []
    for b in range(nblk):
        idx = np.arange(b * 256, m, nblk * 256)
        idx = (idx[:, None] + np.arange(256)[None, :]).ravel(); idx = idx[idx < m]
        blkmin.append(idx[np.argmin(rhs[idx])])
    blkmin = np.array(blkmin)
    tr = [r for _, r in eng.trace()][::-1][:8]
    rows = []
    for r in tr:
        if r not in rows: rows.append(r)
    for r in blkmin[np.argsort(rhs[blkmin], kind="stable")]:
        if len(rows) >= 32: break
        if r not in rows: rows.append(int(r))
    rows = np.array(rows)
    A = T[np.ix_(rows, cj)]
    el = A.astype(f32) <= f32(-1e-6)
    rat = np.where(el, rhs[rows][:, None] / np.where(el, -A, 1.0), np.inf)
    mn = rat.min(axis=0); U = obj[cj] * mn
    fin = np.isfinite(mn)
    k1 = int(np.argmax(np.where(fin, U, -np.inf))); l1 = int(cj[k1])
    a = T[:, l1]; e1 = a.astype(f32) <= f32(-1e-6); r = np.where(e1, rhs / np.where(e1, -a, 1.0), np.inf); S1 = obj[l1] * r.min()
    model = int((np.where(fin, U, np.inf) >= S1).sum())
    exact_rows = np.argsort(rhs, kind="stable")[:32]
    A2 = T[np.ix_(exact_rows, cj)]; el2 = A2.astype(f32) <= f32(-1e-6)
    mn2 = np.where(el2, rhs[exact_rows][:, None] / np.where(el2, -A2, 1.0), np.inf).min(axis=0); U2 = obj[cj] * mn2
    eng.run_pivots(lp.MAX_REDUCTION, 1)
    cur = eng.pricing_stats()
    got = cur[1] - prev[1]; prev = cur
    flag = "   <<<" if got > 200 else ""
    print("pricing %2d gpu %5d model %5d | U[l1] %.6g S1 %.6g argmin row of l1 %d (b %.4g, in R %s, rank of b %d) maxU(exact32) %.6g%s" % (
        it + 1, got, model, U[k1], S1, int(np.argmin(r)), rhs[int(np.argmin(r))], int(np.argmin(r)) in rows.tolist(),
        int((rhs < rhs[int(np.argmin(r))]).sum()), U2.max(), flag))
