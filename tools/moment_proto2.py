#!/usr/bin/env python3
"""Executable specification of the moment-form assembly AS THE GPU DOES IT (avatar_amd/csrc/avt_moments.hip): same
static tables, same storage layouts, same intermediate arrays (X16 per ordered pair, per-(ordered pair, shape key) records,
per-joint partner sums, subtree sums), checked against oracle.evaluate().  See tools/moment_proto.py for the algebra."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avatar_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tools.moment_proto import quat_to_rot  # noqa: E402


def axial(M):   # sum (l x y) given sum l y^T
    return np.array([M[1, 2] - M[2, 1], M[2, 0] - M[0, 2], M[0, 1] - M[1, 0]])


class Tables:
    """What avt_model.cpp::build_moment_tables builds."""
    def __init__(self, smpl, om):
        V, J, K = om.V, om.J, om.K
        self.V, self.J, self.K = V, J, K
        self.S1 = K + 1
        self.NPSI = 3 * self.S1 + 1
        W = np.asarray(smpl["weights"], np.float64)
        self.asg = [sorted([(W[v, j], j) for j in np.nonzero(W[v] > 1e-12)[0]], reverse=True) for v in range(V)]
        self.parent = np.asarray(smpl["kintree_table"])[0].astype(int).copy(); self.parent[0] = -1
        ijp, jsr = om.joint_regression()
        self.jbase, self.jsr = ijp, jsr.reshape(J, 3, K)
        base = np.asarray(smpl["v_template"], np.float64); keys = np.asarray(smpl["shapedirs"], np.float64)
        # psi index a = S1 * i + s (s = 0: base, s >= 1: key s-1); a = 3 S1: the constant 1
        self.psi = np.zeros((V, self.NPSI))
        for i in range(3):
            self.psi[:, self.S1 * i] = base[:, i]
            for s in range(K):
                self.psi[:, self.S1 * i + 1 + s] = keys[:, i, s]
        self.psi[:, 3 * self.S1] = 1.0
        # unordered pairs, k <= k', sorted; per-pair vertex lists (ascending vertex id) with the two weights
        lists = {}
        for v in range(V):
            for wa, a in self.asg[v]:
                for wb, b in self.asg[v]:
                    if a <= b:
                        lists.setdefault((a, b), []).append((v, wa, wb))
        self.pairs = sorted(lists)
        self.lists = [sorted(lists[p]) for p in self.pairs]
        self.NP = len(self.pairs)
        # ordered pairs: op = 2p (k -> k'), 2p+1 (k' -> k; unused for diagonal pairs)
        self.op_first = np.full(2 * self.NP, -1, int); self.op_second = np.full(2 * self.NP, -1, int)
        for p, (a, b) in enumerate(self.pairs):
            self.op_first[2 * p], self.op_second[2 * p] = a, b
            if a != b:
                self.op_first[2 * p + 1], self.op_second[2 * p + 1] = b, a
        self.ops_of = [[op for op in range(2 * self.NP) if self.op_first[op] == k] for k in range(J)]     # ordered pairs whose lever joint is k
        self.sub = [[k for k in range(J) if self.is_under(k, j)] for j in range(J)]                        # subtree of j, ascending
        # rot-rot stage 1: non-empty (k, j') entries: ordered pairs (k, k') with k' under j'
        self.m1 = []            # (k, j', [ops])
        self.m1_of = {}
        for k in range(J):
            for jp in range(J):
                ops = [op for op in self.ops_of[k] if self.is_under(self.op_second[op], jp)]
                if ops:
                    self.m1_of[(k, jp)] = len(self.m1)
                    self.m1.append((k, jp, ops))
        # stage 2: (j <= j' in index order): m1 entries (k, j') with k under j
        self.s2 = []
        for j in range(J):
            for jp in range(j, J):
                ids = [self.m1_of[(k, jp)] for k in self.sub[j] if (k, jp) in self.m1_of]
                ids_t = [self.m1_of[(k, j)] for k in self.sub[jp] if (k, j) in self.m1_of]      # the transposed entry (j', j): its Va is needed
                self.s2.append((j, jp, ids, ids_t))

    def is_under(self, k, j):
        while k >= 0:
            if k == j:
                return True
            k = self.parent[k]
        return False


def k_moments(Tb, cnt, fsum):
    """T[p] full square [NPSI][NPSI]; D[k][NPSI][3]; Efs."""
    T = np.zeros((Tb.NP, Tb.NPSI, Tb.NPSI)); D = np.zeros((Tb.J, Tb.NPSI, 3))
    for p, (a, b) in enumerate(Tb.pairs):
        for v, wa, wb in Tb.lists[p]:
            if cnt[v] > 0:
                T[p] += cnt[v] * wa * wb * np.outer(Tb.psi[v], Tb.psi[v])
                if a == b:
                    D[a] += np.outer(Tb.psi[v], wa * fsum[v])
    m = cnt > 0
    Efs = ((fsum[m] ** 2).sum(1) / cnt[m]).sum()
    return T, D, Efs


def assemble(Tb, T, D, Efs, p, q, w, centre):
    J, K, S1, NPSI, NP = Tb.J, Tb.K, Tb.S1, Tb.NPSI, Tb.NP
    P = 3 + 3 * J + K
    # ---- skeleton tables (what k_solve's skeleton pass leaves in LDS)
    om = np.concatenate([[1.0], w])
    Jpos = Tb.jbase + Tb.jsr @ w
    Rl = [quat_to_rot(q[j]) for j in range(J)]
    R = [None] * J; o = [None] * J; Hs = [None] * J
    for j in range(J):
        pa = Tb.parent[j]
        if pa < 0:
            R[j] = Rl[j]; o[j] = p - centre; Hs[j] = np.zeros((3, K))
        else:
            R[j] = R[pa] @ Rl[j]; o[j] = o[pa] + R[pa] @ (Jpos[j] - Jpos[pa]); Hs[j] = Hs[pa] + R[pa] @ (Tb.jsr[j] - Tb.jsr[pa])
    tau = [o[j] - R[j] @ Jpos[j] for j in range(J)]
    eta = [Hs[j] - R[j] @ Tb.jsr[j] for j in range(J)]
    Rp = [np.eye(3) if Tb.parent[j] < 0 else R[Tb.parent[j]] for j in range(J)]
    # ---- phase A: one 16-lane group per unordered pair, lane = s' (0..K)
    X16 = np.zeros((2 * NP, 16))          # W (9, row-major), Va (3) = sum c a a' x_first, Vb (3) = sum c a a' x_second, t0
    REC = np.zeros((2 * NP, K, 6))        # per (ordered pair, shape key): axial(Y) (3), U (3)
    Zt = np.zeros((K, K)); YX = np.zeros(K)
    for pi, (k, k2) in enumerate(Tb.pairs):
        Tm = T[pi]
        nu = 0.5 if k == k2 else 1.0
        G = R[k].T @ R[k2]
        Ql = np.zeros((S1, 3, 3)); tph = np.zeros((S1, 3)); zz = np.zeros((K, S1))
        for sp in range(S1):                                   # lane s'
            for s in range(S1):
                for i in range(3):
                    for i2 in range(3):
                        v = Tm[S1 * i + s, S1 * i2 + sp]
                        Ql[sp, i, i2] += om[s] * v
                        if s >= 1:
                            zz[s - 1, sp] += G[i, i2] * v
            tph[sp] = [Tm[NPSI - 1, S1 * i2 + sp] for i2 in range(3)]
        t0 = Tm[NPSI - 1, NPSI - 1]
        P2 = np.einsum("s,sij->ij", om, Ql)                    # group reduction over the lanes
        p1 = om @ tph
        for a, b, op, tr in ((k, k2, 2 * pi, False),) + (((k2, k, 2 * pi + 1, True),) if k != k2 else ()):
            P2ab = P2.T if tr else P2
            Wm = R[a] @ P2ab @ R[b].T + np.outer(R[a] @ p1, tau[b]) + np.outer(tau[a], R[b] @ p1) + t0 * np.outer(tau[a], tau[b])
            Va = R[a] @ p1 + t0 * tau[a]
            Vb = R[b] @ p1 + t0 * tau[b]
            X16[op, :9] = Wm.reshape(-1); X16[op, 9:12] = Va; X16[op, 12:15] = Vb; X16[op, 15] = t0
            for sp in range(1, S1):                            # lane s' >= 1: shape key s' - 1
                s = sp - 1
                Qs = Ql[sp]                                    # sum c a a' (Phi om)_i (Phi e_s)_i': the same array whichever joint carries which factor
                y_lin = R[b] @ tph[sp]
                Y = R[a] @ Qs @ R[b].T + np.outer(R[a] @ p1, eta[b][:, s]) + np.outer(tau[a], y_lin) + t0 * np.outer(tau[a], eta[b][:, s])
                U = y_lin + t0 * eta[b][:, s]
                REC[op, s, :3] = axial(Y); REC[op, s, 3:] = U
                YX[s] += np.trace(Y)
        for sp in range(1, S1):                                # lane-local column t = sp - 1 of Z~'
            t = sp - 1
            for s in range(K):
                Zt[s, t] += nu * (zz[s, sp] + eta[k][:, s] @ (R[k2] @ tph[sp]) + eta[k2][:, s] @ (R[k] @ tph[sp]) + t0 * (eta[k][:, s] @ eta[k2][:, t]))
    # ---- phase A': per joint, the data-side moments
    XD = np.zeros((J, 3, 3)); Dl = np.zeros((J, 3)); YF = np.zeros(K)
    for k in range(J):
        Dphi = D[k][:NPSI - 1].reshape(3, S1, 3)               # [i][s][c]
        Dl[k] = D[k][NPSI - 1]
        XD[k] = R[k] @ np.einsum("s,isc->ic", om, Dphi) + np.outer(tau[k], Dl[k])
        for s in range(K):
            YF[s] += np.einsum("ci,ic->", R[k], Dphi[:, s + 1, :]) + eta[k][:, s] @ Dl[k]
    # ---- phase B0: per-joint partner sums
    PK = np.zeros((J, 16)); PR = np.zeros((J, K, 6))           # PK: axial(sum W) 3, sum Va 3, sum Vb 3, sum t0, axial(XD) 3, Dl 3
    for k in range(J):
        for op in Tb.ops_of[k]:
            PK[k, :3] += axial(X16[op, :9].reshape(3, 3)); PK[k, 3:6] += X16[op, 9:12]; PK[k, 6:9] += X16[op, 12:15]; PK[k, 9] += X16[op, 15]
            PR[k] += REC[op]
        PK[k, 10:13] = axial(XD[k]); PK[k, 13:16] = Dl[k]
    # ---- phase B1: subtree sums
    TK = np.zeros((J, 16)); TR = np.zeros((J, K, 6))
    for j in range(J):
        for k in Tb.sub[j]:
            TK[j] += PK[k]; TR[j] += PR[k]
    H = np.zeros((P + 1, P + 1))      # row / column P: J^T r, [P][P]: sum c |r|^2
    # ---- phase B2: everything but rot-rot
    ctot = TK[0, 9]
    for c in range(3):
        H[c, c] = ctot
    g_tr = TK[0, 3:6] - TK[0, 13:16]
    H[:3, P] = g_tr; H[P, :3] = g_tr
    for j in range(J):
        lam = TK[j, 3:6] - TK[j, 9] * o[j]
        lr = (TK[j, :3] - np.cross(o[j], TK[j, 6:9])) - (TK[j, 10:13] - np.cross(o[j], TK[j, 13:16]))
        for c in range(3):
            a = Rp[j][:, c]
            col = 2.0 * np.cross(a, lam)
            r = 3 + 3 * j + c
            H[r, :3] = col; H[:3, r] = col
            H[r, P] = H[P, r] = 2.0 * a @ lr
            for s in range(K):
                v = 2.0 * a @ (TR[j, s, :3] - np.cross(o[j], TR[j, s, 3:]))
                H[r, 3 + 3 * J + s] = H[3 + 3 * J + s, r] = v
    for s in range(K):
        H[3 + 3 * J + s, :3] = TR[0, s, 3:]; H[:3, 3 + 3 * J + s] = TR[0, s, 3:]
        H[3 + 3 * J + s, P] = H[P, 3 + 3 * J + s] = YX[s] - YF[s]
    H[3 + 3 * J:P, 3 + 3 * J:P] = Zt + Zt.T
    xx = sum(np.trace(X16[op, :9].reshape(3, 3)) for op in range(2 * NP))
    xf = sum(np.trace(XD[k]) for k in range(J))
    H[P, P] = xx - 2.0 * xf + Efs
    # ---- rot-rot: stage 1 (k, j'), stage 2 (j <= j'), stage 3 blocks
    M1 = np.zeros((len(Tb.m1), 16))
    for e, (k, jp, ops) in enumerate(Tb.m1):
        for op in ops:
            M1[e] += X16[op]
    for (j, jp, ids, ids_t) in Tb.s2:
        S = np.zeros(16)
        for e in ids:
            S += M1[e]
        St = np.zeros(16)
        for e in ids_t:
            St += M1[e]
        # LL[j,j'] = SW - SVa o_j'^T - o_j SVb'^T + ST o_j o_j'^T with SVb' = sum V_{k'k} = (transposed entry).Va
        LL = S[:9].reshape(3, 3) - np.outer(S[9:12], o[jp]) - np.outer(o[j], St[9:12]) + S[15] * np.outer(o[j], o[jp])
        blk = 4.0 * (np.trace(LL) * Rp[j].T @ Rp[jp] - Rp[j].T @ LL.T @ Rp[jp])
        H[3 + 3 * j:6 + 3 * j, 3 + 3 * jp:6 + 3 * jp] = blk
        H[3 + 3 * jp:6 + 3 * jp, 3 + 3 * j:6 + 3 * j] = blk.T
    return H


def main():
    smpl = synth.load_model(0)
    om = orc.OracleModel(smpl)
    Tb = Tables(smpl, om)
    nlist = [len(l) for l in Tb.lists]
    print(f"pairs {Tb.NP}, list entries {sum(nlist)} (max {max(nlist)} for pair {Tb.pairs[int(np.argmax(nlist))]}), m1 entries {len(Tb.m1)} "
          f"(ops total {sum(len(x[2]) for x in Tb.m1)}), stage-2 lists {sum(len(x[2]) + len(x[3]) for x in Tb.s2)} over {len(Tb.s2)} blocks, "
          f"max ops per joint {max(len(x) for x in Tb.ops_of)}")
    pm = synth.identity_part_map()
    for seed, dense in ((0, False), (1, False), (2, False), (0, True)):
        fr = synth.make_frame(smpl, seed, dense=dense)
        w0, p0, R0 = fr["start"]
        q0 = orc.rot_to_quat(R0)
        cloud, _, _ = om.update(w0, p0, R0)
        vis = om.visibility(cloud)
        corr = om.nn(pm, 24, cloud, vis, fr["data"], fr["labels"])
        data = np.asarray(fr["data"], np.float64).reshape(-1, 3)
        centre = data[0].copy()                              # avt_bucket.h: the frame's first data point
        cnt = np.bincount(corr[corr >= 0], minlength=Tb.V).astype(float)
        fsum = np.zeros((Tb.V, 3)); np.add.at(fsum, corr[corr >= 0], data[corr >= 0] - centre)
        cc = np.zeros(Tb.V); np.add.at(cc, corr[corr >= 0], ((data[corr >= 0] - centre) ** 2).sum(1))
        m = cnt > 0
        cost_const = 0.5 * (cc[m] - (fsum[m] ** 2).sum(1) / cnt[m]).sum()
        T, D, Efs = k_moments(Tb, cnt, fsum)
        mpl = [sum(1 for v, _, _ in l if cnt[v] > 0) for l in Tb.lists]
        c_, g_, H_, _ = om.evaluate(p0, q0, w0, corr, data, 0.0, 0.0, aggregate=1)
        delta = -np.linalg.solve(H_ + 1e-3 * np.diag(np.diag(H_)) + 1e-9 * np.eye(len(g_)), g_)
        for (p, q, w) in ((p0, q0, w0), om.retract(p0, q0, w0, delta)):
            cost_o, g_o, H_o, _ = om.evaluate(p, q, w, corr, data, 0.0, 0.0, aggregate=1)
            H = assemble(Tb, T, D, Efs, np.asarray(p), np.asarray(q).reshape(-1, 4), np.asarray(w), centre)
            P = len(g_o)
            eH = np.abs(H[:P, :P] - H_o).max() / np.abs(H_o).max(); eg = np.abs(H[:P, P] - g_o).max() / np.abs(g_o).max()
            ec = abs(0.5 * H[P, P] + cost_const - cost_o) / cost_o
            print(f"seed {seed} dense {dense}: matched {int(m.sum())}, longest matched list {max(mpl)}; H {eH:.2e} g {eg:.2e} cost {ec:.2e}")
            assert eH < 1e-10 and eg < 1e-10 and ec < 1e-10


if __name__ == "__main__":
    main()
