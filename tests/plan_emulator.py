"""Test helper: executes a bt_plan on the CPU in float64 numpy, mirroring what the
HIP kernels do with the plan arrays (ba_kernels.hip: k_prep, k_tile,
k_pair_finalize, k_solve, k_update), with the per-edge Jacobians taken from the
oracle.  It validates the plan layout, the Ji = -Jj Ad algebra and the
block-sparse factorisation structure without a GPU.  Test infrastructure only.
"""
import numpy as np

import oracle


def quat_rot_matrix(q):
    q = q / np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)],
                     [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                     [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])


def hat(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0.0]])


def pair_geometry(poses, i, j):
    if i == j:
        return np.eye(3), np.zeros(3)
    Ri, Rj = quat_rot_matrix(poses[i, 3:]), quat_rot_matrix(poses[j, 3:])
    R = Rj @ Ri.T
    return R, poses[j, :3] - R @ poses[i, :3]


def adjoint(R, t):
    Ad = np.zeros((6, 6))
    Ad[:3, :3] = R
    Ad[:3, 3:] = hat(t) @ R
    Ad[3:, 3:] = R
    return Ad


def run(plan, A, inp, wkey, lmbda=1e-4, ep=10.0, alpha=0.05, structure_only=False, loss="huber"):
    """plan: batrack_amd.plan.Plan (host-only is fine); A = plan.arrays(); inp: dict of
    float64 inputs (poses, patches, mono, intrinsics, targets3, weights.., ii, jj, kk, bounds)."""
    info = plan.info
    n, fixedp, D = info["n"], info["fixedp"], 6 * info["n"]
    ed = oracle.edges(inp["poses"], inp["patches"], inp["intrinsics"], inp["targets3"], inp[wkey],
                      inp["ii"], inp["jj"], inp["kk"], inp["bounds"], loss=loss)
    Jj, Jz, r, W = ed["Jj"], ed["Jz"], ed["r"], ed["W"]
    P = info["pairs"]
    Ads = [adjoint(*pair_geometry(inp["poses"], int(A["pair_i"][p]), int(A["pair_j"][p]))) for p in range(P)]
    S = np.zeros((D, D))
    y = np.zeros(D)
    Bjj = np.zeros((P, 6, 6))
    gj = np.zeros((P, 6))
    m = info["m"]
    Q = np.zeros(m)
    wp = np.zeros(m)
    Esave = {}
    so = structure_only or n == 0
    for t in range(info["tiles"]):
        nt, nc = int(A["tile_ntrk"][t]), int(A["tile_ncam"][t])
        cams = A["tile_cams"][A["tile_cam0"][t]:A["tile_cam0"][t] + nc]
        Eh = np.zeros((6 * nc + 1, 64))
        C = np.zeros(64)
        wv = np.zeros(64)
        for s in range(int(A["tile_nslot"][t])):
            base = (int(A["tile_slot0"][t]) + s) * 64
            for lane in range(64):
                e = int(A["slot_edge"][base + lane])
                if e < 0:
                    continue
                assert lane < nt
                p = int(A["slot_pair"][base + lane])
                lab = int(A["slot_lab"][base + lane])
                la, lb = lab & 0xff, lab >> 8
                assert A["pair_i"][p] == inp["ii"][e] and A["pair_j"][p] == inp["jj"][e]
                assert A["kx"][A["tile_trk0"][t] + lane] == inp["kk"][e]
                Wd = np.diag(W[e])
                C[lane] += Jz[e] @ Wd @ Jz[e]
                wv[lane] += Jz[e] @ Wd @ r[e]
                Ej = Jj[e].T @ Wd @ Jz[e]
                if lb != 0xff:
                    assert cams[lb] == inp["jj"][e] - fixedp
                    Eh[6*lb:6*lb + 6, lane] += Ej
                else:
                    assert inp["jj"][e] < fixedp
                if la != 0xff:
                    assert cams[la] == inp["ii"][e] - fixedp
                    Eh[6*la:6*la + 6, lane] += -Ads[p].T @ Ej
                else:
                    assert inp["ii"][e] < fixedp
                Bjj[p] += Jj[e].T @ Wd @ Jj[e]
                gj[p] += Jj[e].T @ Wd @ r[e]
        for lane in range(nt):
            trk = int(A["tile_trk0"][t]) + lane
            patch = int(A["kx"][trk])
            mono = inp["mono"][patch]
            pm = 1.0 if mono > 1e-2 else 0.0
            Q[trk] = 1.0 / (C[lane] + pm * alpha + lmbda)
            wp[trk] = wv[lane] - pm * alpha * (inp["patches"][patch, 2] - mono)
            Eh[6 * nc, lane] = wp[trk]
        if so:
            continue
        Esave[t] = Eh[:6 * nc].copy()
        Ql = np.zeros(64)
        Ql[:nt] = Q[A["tile_trk0"][t]:A["tile_trk0"][t] + nt]
        out = (Eh * Ql) @ Eh.T
        gidx = np.array([6 * cams[c // 6] + c % 6 for c in range(6 * nc)], dtype=np.int64)
        for rr in range(6 * nc):
            for cc in range(6 * nc):
                if gidx[rr] >= gidx[cc]:
                    S[gidx[rr], gidx[cc]] -= out[rr, cc]
        y[gidx] -= out[6 * nc, :6 * nc]
    if not so:
        for p in range(P):
            a, b = int(A["pair_i"][p]) - fixedp, int(A["pair_j"][p]) - fixedp
            Ad = Ads[p]
            M = Bjj[p] @ Ad
            low = np.tril(np.ones((6, 6), bool))
            if a >= 0:
                blk = Ad.T @ M
                S[6*a:6*a + 6, 6*a:6*a + 6] += np.where(low, blk, 0)
                y[6*a:6*a + 6] += -Ad.T @ gj[p]
            if b >= 0:
                S[6*b:6*b + 6, 6*b:6*b + 6] += np.where(low, Bjj[p], 0)
                y[6*b:6*b + 6] += gj[p]
            if a >= 0 and b >= 0:
                if a > b:
                    S[6*a:6*a + 6, 6*b:6*b + 6] += -M.T
                elif b > a:
                    S[6*b:6*b + 6, 6*a:6*a + 6] += -M
                else:
                    S[6*a:6*a + 6, 6*a:6*a + 6] += np.where(low, -(M + M.T), 0)
    out = dict(S_lower=S, y=y, Q=Q, wp=wp)
    dX = np.zeros((n, 6))
    if not so:
        dX = sparse_chol_solve(A, S, y, n, ep, 1e-4)
        dX2 = sparse_chol_solve_fused(A, S, y, n, ep, 1e-4)
        assert np.allclose(dX, dX2, rtol=1e-9, atol=1e-12 * max(1.0, np.abs(dX).max())), "fused schedule differs from the two-phase schedule"
        out["dX"] = dX
    # update
    patches_out = inp["patches"].copy()
    dz_all = np.zeros(inp["patches"].shape[0])
    for t in range(info["tiles"]):
        nt, nc = int(A["tile_ntrk"][t]), int(A["tile_ncam"][t])
        cams = A["tile_cams"][A["tile_cam0"][t]:A["tile_cam0"][t] + nc]
        for lane in range(nt):
            trk = int(A["tile_trk0"][t]) + lane
            assert A["trk_loc"][trk] == (t << 6 | lane)
            assert A["trk_of_patch"][A["kx"][trk]] == trk
            if so:
                dz = Q[trk] * wp[trk]
            else:
                acc = sum(Esave[t][6*c:6*c + 6, lane] @ dX[cams[c]] for c in range(nc))
                dz = Q[trk] * (wp[trk] - acc)
            dz_all[A["kx"][trk]] = dz
    patches_out[:, 2] = np.clip(inp["patches"][:, 2] + dz_all, 1e-3, 10.0)
    out["patches_out"] = patches_out
    return out


def sparse_chol_solve(A, S_lower, y, n, ep, lm):
    """k_solve_lds: level-scheduled block-sparse right-looking Cholesky driven by the plan
    arrays (perm, blk_src, col_ptr/row_idx, upd/upd_next, lvl_ptr/lvl_cols, dp)."""
    col_ptr, row_idx, upd_ptr, upd_next = A["col_ptr"], A["row_idx"], A["upd_ptr"], A["upd_next"]
    upd = A["upd"].reshape(-1, 3)
    perm, blk_src, blk_col, yshared = A["perm"], A["blk_src"], A["blk_col"] & 255, A["blk_col"] >> 16
    lvl_ptr, lvl_cols, col_lvl, dp_ptr, dp = A["lvl_ptr"], A["lvl_cols"], A["col_lvl"], A["dp_ptr"], A["dp"]
    nnzb, nlev = len(row_idx), len(lvl_ptr) - 1
    assert sorted(perm.tolist()) == list(range(n)) and sorted(lvl_cols.tolist()) == list(range(n))
    # every structurally non-zero block of S must be inside the symbolic pattern (in permuted numbering)
    Sfull = S_lower + np.tril(S_lower, -1).T
    Sp = Sfull.reshape(n, 6, n, 6)[perm][:, :, perm]            # [n,6,n,6] permuted
    Sb = np.abs(Sp).max(axis=(1, 3))
    pat = np.zeros((n, n), bool)
    pat[row_idx, blk_col] = True
    assert not np.any(np.tril(Sb > 0) & ~pat), "S has a block outside the plan's sparsity pattern"
    L = np.zeros((nnzb, 6, 6))
    for b in range(nnzb):
        src = int(blk_src[b]); rn, cn, tr = src >> 9, (src >> 1) & 255, src & 1
        i, j = int(row_idx[b]), int(blk_col[b])
        assert {rn, cn} == {int(perm[i]), int(perm[j])} and rn >= cn
        blk = S_lower[6*rn:6*rn + 6, 6*cn:6*cn + 6].copy()
        if i == j:
            blk = np.tril(blk)
            blk[np.diag_indices(6)] += ep + lm * np.diag(blk)
        elif tr:
            blk = blk.T.copy()
        assert np.allclose(blk if i != j else blk + np.tril(blk, -1).T, Sp[i, :, j, :] + (np.eye(6) * (ep + lm * np.diag(Sp[i, :, j, :])) if i == j else 0))
        L[b] = blk
    z = y.reshape(n, 6)[perm].reshape(-1).copy()
    Linv = np.zeros((n, 6, 6))
    applied = np.zeros(len(upd), int)
    for l in range(nlev):
        cols = lvl_cols[lvl_ptr[l]:lvl_ptr[l + 1]]
        assert 1 <= len(cols) <= 4 and all(col_lvl[c] == l for c in cols)
        prev = lvl_cols[lvl_ptr[l - 1]:lvl_ptr[l]] if l > 0 else []
        # phase 1, helper side: all but the leading `upd_next` triples of the previous level's columns
        targets = {}
        for p in prev:
            for t in range(upd_ptr[p] + upd_next[p], upd_ptr[p + 1]):
                s1, s2, dstf = upd[t]
                dst, flag = int(dstf) & 0x7fff, int(dstf) >> 15
                assert blk_col[s1] == p and blk_col[s2] == p
                assert row_idx[dst] == row_idx[s1] and blk_col[dst] == row_idx[s2]
                targets.setdefault(dst, set()).add(int(p))
                L[dst] -= L[s1] @ L[s2].T
                applied[t] += 1
        for p in prev:
            for t in range(upd_ptr[p] + upd_next[p], upd_ptr[p + 1]):
                dstf = int(upd[t][2])
                assert (dstf >> 15) == (1 if len(targets[dstf & 0x7fff]) > 1 else 0), "shared-destination flag wrong"
            for s in range(col_ptr[p] + 1, col_ptr[p + 1]):
                i = int(row_idx[s])
                n_writers = sum(1 for p2 in prev if i in row_idx[col_ptr[p2] + 1:col_ptr[p2 + 1]])
                assert yshared[s] == (1 if n_writers > 1 else 0), "shared-y flag wrong"
                z[6*i:6*i + 6] -= L[s] @ z[6*p:6*p + 6]
        # phase 1, critical side: pending diagonal updates, then the 6x6 factorisation
        for c in cols:
            for t in dp[dp_ptr[c]:dp_ptr[c + 1]]:
                s1, s2, dstf = upd[t]
                assert s1 == s2 and (int(dstf) & 0x7fff) == col_ptr[c] and col_lvl[blk_col[s1]] == l - 1
                assert upd_ptr[blk_col[s1]] <= t < upd_ptr[blk_col[s1]] + upd_next[blk_col[s1]]
                L[col_ptr[c]] -= L[s1] @ L[s2].T
                applied[t] += 1
            d = col_ptr[c]
            full = L[d] + np.tril(L[d], -1).T
            Lj = np.linalg.cholesky(full)
            L[d] = Lj
            Linv[c] = np.linalg.inv(Lj)
        # phase 2: block rows and the RHS row
        for c in cols:
            z[6*c:6*c + 6] = Linv[c] @ z[6*c:6*c + 6]
            for s in range(col_ptr[c] + 1, col_ptr[c + 1]):
                assert col_lvl[row_idx[s]] > l
                L[s] = L[s] @ Linv[c].T
    assert np.all(applied == 1), "every update triple must be applied exactly once"
    x = z.copy()
    for l in range(nlev - 1, -1, -1):
        for c in lvl_cols[lvl_ptr[l]:lvl_ptr[l + 1]]:
            tq = x[6*c:6*c + 6].copy()
            for s in range(col_ptr[c] + 1, col_ptr[c + 1]):
                i = int(row_idx[s])
                tq -= L[s].T @ x[6*i:6*i + 6]
            x[6*c:6*c + 6] = Linv[c].T @ tq
    out = np.zeros((n, 6))
    out[perm] = x.reshape(n, 6)
    return out


def sparse_chol_solve_fused(A, S_lower, y, n, ep, lm):
    """k_solve_fused: one phase per level.  The waves of a column apply the pending updates (sources
    one level below) to the column's own blocks, factor and substitute; the lazy updates of the level
    below run concurrently on the helper waves.  Checks that the two groups touch disjoint data."""
    col_ptr, row_idx = A["col_ptr"], A["row_idx"]
    perm, blk_src, blk_col, yshared = A["perm"], A["blk_src"], A["blk_col"] & 255, A["blk_col"] >> 16
    lvl_ptr, lvl_cols, col_lvl = A["lvl_ptr"], A["lvl_cols"], A["col_lvl"]
    pend_ptr, pend = A["fz_pend_ptr"], A["fz_pend"].reshape(-1, 2)
    lazy_ptr, lazy, yurg = A["fz_lazy_ptr"], A["fz_lazy"].reshape(-1, 3), A["fz_yurg"]
    meta = A["fz_meta"].reshape(-1, 4, 8)
    nnzb, nlev = len(row_idx), len(lvl_ptr) - 1
    ri, pf = A["fz_rowinfo"], A["fz_pfirst"].astype(np.int64) & 0xffffffff
    for b in range(nnzb):
        assert ri[b] == (int(row_idx[b]) | int(blk_col[b]) << 8 | int(yshared[b]) << 24 | int(yurg[b]) << 25)
        c = int(pend_ptr[b + 1] - pend_ptr[b])
        want = (int(pend[pend_ptr[b]][0]) | int(pend[pend_ptr[b]][1]) << 15 | min(c, 3) << 30) if c else 0
        assert pf[b] == want
        want2 = (int(pend[pend_ptr[b] + 1][0]) | int(pend[pend_ptr[b] + 1][1]) << 15) if c > 1 else 0
        assert A["fz_psecond"][b] == want2 and c <= 2
    assert len(pend) + len(lazy) == len(A["upd"]) // 3, "every update triple is either pending or lazy"
    L = np.zeros((nnzb, 6, 6))
    for b in range(nnzb):
        src = int(blk_src[b]); rn, cn, tr = src >> 9, (src >> 1) & 255, src & 1
        blk = S_lower[6*rn:6*rn + 6, 6*cn:6*cn + 6].copy()
        if row_idx[b] == blk_col[b]:
            blk = np.tril(blk)
            blk[np.diag_indices(6)] += ep + lm * np.diag(blk)
        elif tr:
            blk = blk.T.copy()
        L[b] = blk
    z = y.reshape(n, 6)[perm].reshape(-1).copy()
    Linv = np.zeros((n, 6, 6))
    for l in range(nlev):
        cols = [int(c) for c in lvl_cols[lvl_ptr[l]:lvl_ptr[l + 1]]]
        prev = [int(c) for c in lvl_cols[lvl_ptr[l - 1]:lvl_ptr[l]]] if l > 0 else []
        w0 = 0
        for q in range(4):
            m = meta[l, q]
            if q < len(cols):
                c = cols[q]
                assert m[0] == c and m[1] == col_ptr[c] and m[2] == col_ptr[c + 1] - col_ptr[c] - 1 and m[7] == len(cols)
                assert m[3] == lazy_ptr[c] and m[4] == lazy_ptr[c + 1] - lazy_ptr[c]
                assert m[5] == w0 and m[6] == (6 * m[2] + 1 + 63) // 64
                w0 += int(m[6])
            else:
                assert m[0] == -1
        pm = A["fz_pmeta"].reshape(-1, 8)[l]
        assert pm[1] >> 24 == len(cols)
        for q, c in enumerate(cols[:2]):
            d = int(col_ptr[c]); cnt = int(col_ptr[c + 1]) - d - 1
            npd = int(pend_ptr[d + 1] - pend_ptr[d])
            sd = int(pend[pend_ptr[d]][0]) if npd else 0
            ysrc = int(blk_col[sd]) if npd else 0
            assert pm[4*q] == (c | cnt << 8 | ysrc << 16) and (pm[4*q + 1] & 0xffffff) == (d | ((6 * cnt + 1 + 63) // 64) << 16)
            assert pm[4*q + 2] == (int(lazy_ptr[c]) | (int(lazy_ptr[c + 1] - lazy_ptr[c]) << 16))
            assert (pm[4*q + 3] & 0xfffff) == (sd | min(npd, 3) << 15)
        a_reads, a_writes = set(), set()              # blocks; ("z", col) for y segments
        for c in cols:
            d = int(col_ptr[c])
            for b in range(d, int(col_ptr[c + 1])):
                for s1, s2 in pend[pend_ptr[b]:pend_ptr[b + 1]]:
                    src_col = int(blk_col[s1])
                    assert blk_col[s2] == src_col and col_lvl[src_col] == l - 1
                    assert row_idx[s1] == row_idx[b] and row_idx[s2] == c
                    L[b] -= L[s1] @ L[s2].T
                    a_reads |= {int(s1), int(s2)}
                a_writes.add(b)
            for s1, s2 in pend[pend_ptr[d]:pend_ptr[d + 1]]:          # pending y contributions ride on the diagonal list
                assert s1 == s2 and yurg[s1] == 1
                src_col = int(blk_col[s1])
                z[6*c:6*c + 6] -= L[s1] @ z[6*src_col:6*src_col + 6]
                a_reads.add(("z", src_col))
            n_yurg = sum(int(yurg[b]) for p in prev for b in range(col_ptr[p] + 1, col_ptr[p + 1]) if row_idx[b] == c)
            assert n_yurg == pend_ptr[d + 1] - pend_ptr[d]
            a_writes.add(("z", c))
            full = np.tril(L[d]) + np.tril(L[d], -1).T
            Lj = np.linalg.cholesky(full)
            L[d] = Lj
            Linv[c] = np.linalg.inv(Lj)
            z[6*c:6*c + 6] = Linv[c] @ z[6*c:6*c + 6]
            for s in range(d + 1, col_ptr[c + 1]):
                assert col_lvl[row_idx[s]] > l
                L[s] = L[s] @ Linv[c].T
        # lazy updates of the level below, on the helper waves during the same phase
        targets, ytargets = {}, {}
        for p in prev:
            for s1, s2, dstf in lazy[lazy_ptr[p]:lazy_ptr[p + 1]]:
                dst = int(dstf) & 0x7fff
                assert blk_col[s1] == p and blk_col[s2] == p and row_idx[dst] == row_idx[s1] and blk_col[dst] == row_idx[s2]
                assert col_lvl[blk_col[dst]] > l, "a lazy destination must lie above the level being factored"
                assert dst not in a_reads and dst not in a_writes and int(s1) not in a_writes and int(s2) not in a_writes
                targets.setdefault(dst, []).append((p, int(dstf) >> 15))
                L[dst] -= L[s1] @ L[s2].T
            for b in range(col_ptr[p] + 1, col_ptr[p + 1]):
                if yurg[b]:
                    assert col_lvl[row_idx[b]] == l
                    continue
                i = int(row_idx[b])
                assert col_lvl[i] > l and ("z", i) not in a_reads and ("z", i) not in a_writes and ("z", p) not in a_writes
                ytargets.setdefault(i, []).append(int(yshared[b]))
                z[6*i:6*i + 6] -= L[b] @ z[6*p:6*p + 6]
        for dst, ws in targets.items():
            if len(ws) > 1:
                assert all(f == 1 for _, f in ws), "two columns update a block without the shared flag"
        for i, fl in ytargets.items():
            if len(fl) > 1:
                assert all(f == 1 for f in fl), "two columns update a y segment without the shared flag"
    # back substitution: one wave per column slot, barriers only where bs_sync says so; a value written by
    # another slot's wave may be read only if a barrier lies in between
    x = z.copy()
    bs_sync = A["bs_sync"]
    written_at = {}                          # column -> (slot, number of barriers passed when written)
    barriers = 0
    for l in range(nlev - 1, -1, -1):
        if bs_sync[l]:
            barriers += 1
        for q, c in enumerate(lvl_cols[lvl_ptr[l]:lvl_ptr[l + 1]]):
            tq = x[6*c:6*c + 6].copy()
            for s in range(col_ptr[c] + 1, col_ptr[c + 1]):
                i = int(row_idx[s])
                sl, nb = written_at[i]
                assert sl == q or nb < barriers, "back substitution reads another wave's x without a barrier"
                tq -= L[s].T @ x[6*i:6*i + 6]
            x[6*c:6*c + 6] = Linv[c].T @ tq
            written_at[int(c)] = (q, barriers)
    out = np.zeros((n, 6))
    out[perm] = x.reshape(n, 6)
    return out


def pipe_schedule_applies(A):
    """The plan-side condition of k_solve_pipe (ba_plan.cpp: fzp_ok)."""
    col_ptr, lvl_ptr = A["col_ptr"], A["lvl_ptr"]
    n = len(col_ptr) - 1
    return (n > 0 and np.diff(lvl_ptr).max() <= 2 and np.diff(A["fz_pend_ptr"]).max(initial=0) <= 2 and
            all(6 * (col_ptr[j + 1] - col_ptr[j] - 1) + 1 <= 64 for j in range(n)))


def check_pipe_protocol(A, nw=12):
    """k_solve_pipe has no workgroup barrier in the sweep: waves with fixed roles are ordered by the flags
    lready / colready / hcnt only.  Build the happens-before relation those waits imply (program order per
    wave + flag edges) and verify that EVERY pair of steps touching the same data with at least one write is
    ordered by it.  Steps: D(l, q) diagonal wave, R(l, q) row wave, H(b, w) helper wave w on lazy batch b."""
    col_ptr, row_idx = A["col_ptr"], A["row_idx"]
    blk_col = A["blk_col"] & 255
    lvl_ptr, lvl_cols = A["lvl_ptr"], A["lvl_cols"]
    pend_ptr, pend = A["fz_pend_ptr"], A["fz_pend"].reshape(-1, 2)
    lazy_ptr, lazy, yurg = A["fz_lazy_ptr"], A["fz_lazy"].reshape(-1, 3), A["fz_yurg"]
    nlev = len(lvl_ptr) - 1
    nh, hs = nw - 4, (nw - 4) * 64
    cols = [[int(c) for c in lvl_cols[lvl_ptr[l]:lvl_ptr[l + 1]]] for l in range(nlev)]
    nodes, idx = [], {}

    def add(key, reads, writes):
        idx[key] = len(nodes)
        nodes.append((key, reads, writes))

    ROWS = range(6)
    for l in range(nlev):
        for q, c in enumerate(cols[l]):
            d = int(col_ptr[c])
            srcs = {(int(s1), r) for s1, _ in pend[pend_ptr[d]:pend_ptr[d + 1]] for r in ROWS}
            add(("D", l, q), {(d, r) for r in ROWS} | srcs, {(d, r) for r in ROWS} | {("scr", q)})
            reads, writes = {("scr", q), ("z", c)}, {("z", c)}
            for b in range(d + 1, int(col_ptr[c + 1])):
                writes |= {(b, r) for r in ROWS}
                reads |= {(b, r) for r in ROWS}
                for s1, s2 in pend[pend_ptr[b]:pend_ptr[b + 1]]:
                    reads |= {(int(s1), r) for r in ROWS} | {(int(s2), r) for r in ROWS}
            for s1, _ in pend[pend_ptr[d]:pend_ptr[d + 1]]:                  # pending y contributions
                reads |= {(int(s1), r) for r in ROWS} | {("z", int(blk_col[s1]))}
            add(("R", l, q), reads, writes)
    for b in range(nlev - 1):
        per_wave = [(set(), set()) for _ in range(nh)]
        item = 0
        for c in cols[b]:
            for s1, s2, dstf in lazy[lazy_ptr[c]:lazy_ptr[c + 1]]:
                for r in ROWS:
                    w = (item % hs) // 64
                    per_wave[w][0].update({(int(s1), r)} | {(int(s2), rr) for rr in ROWS} | {(int(dstf) & 0x7fff, r)})
                    per_wave[w][1].add((int(dstf) & 0x7fff, r))
                    item += 1
        shift = ((item + 63) // 64) * 64
        yi = 0
        for c in cols[b]:
            for blk in range(int(col_ptr[c]) + 1, int(col_ptr[c + 1])):
                for r in ROWS:
                    if not yurg[blk]:
                        w = ((shift + yi) % hs) // 64
                        per_wave[w][0].update({(blk, r), ("z", c), ("z", int(row_idx[blk]))})
                        per_wave[w][1].add(("z", int(row_idx[blk])))
                    yi += 1
        for w in range(nh):
            add(("H", b, w), per_wave[w][0], per_wave[w][1])
    N = len(nodes)
    hb = np.zeros((N, N), bool)

    def edge(a, b):
        if a in idx and b in idx:
            hb[idx[a], idx[b]] = True

    for q in range(2):                                           # program order of the column waves
        for kind in "DR":
            seq = [(kind, l, q) for l in range(nlev) if q < len(cols[l])]
            for x, y in zip(seq, seq[1:]):
                edge(x, y)
    for w in range(nh):
        for b in range(nlev - 2):
            edge(("H", b, w), ("H", b + 1, w))
    for l in range(nlev):
        for q in range(len(cols[l])):
            edge(("D", l, q), ("R", l, q))                       # lready
            if l > 0:
                dep = (int(A["fz_pmeta"].reshape(-1, 8)[l][4 * q + 3]) >> 20) & 3
                for q2 in range(len(cols[l - 1])):               # colready of the columns that hold pending sources
                    if dep >> q2 & 1:
                        edge(("R", l - 1, q2), ("D", l, q)); edge(("R", l - 1, q2), ("R", l, q))
                if q < len(cols[l - 1]):                         # the diagonal wave's scratch slot: its last reader
                    edge(("R", l - 1, q), ("D", l, q))
            if l >= 2:
                for w in range(nh):                              # hcnt >= nh (l - 1): batches 0 .. l - 2 complete
                    edge(("H", l - 2, w), ("D", l, q)); edge(("H", l - 2, w), ("R", l, q))
    for b in range(nlev - 1):
        for w in range(nh):
            for q2 in range(len(cols[b])):
                edge(("R", b, q2), ("H", b, w))                  # sources of the batch
            if b > 0:
                for w2 in range(nh):
                    edge(("H", b - 1, w2), ("H", b, w))          # hcnt >= nh b: the group has finished the batches before
    for k in range(N):                                           # transitive closure (Warshall, vectorised)
        hb |= np.outer(hb[:, k], hb[k, :])
    bad = []
    for i in range(N):
        ki, ri, wi = nodes[i]
        for j in range(i + 1, N):
            kj, rj, wj = nodes[j]
            if ki[0] == "H" and kj[0] == "H" and ki[1] == kj[1]:
                continue                                         # same batch: shared destinations carry the atomic flag (checked elsewhere)
            if (wi & (rj | wj)) or (wj & ri):
                if not (hb[i, j] or hb[j, i]):
                    bad.append((ki, kj, sorted((wi & (rj | wj)) | (wj & ri), key=str)[:3]))
    assert not bad, f"unordered conflicting steps, e.g. {bad[:3]}"
    return N
