"""Test infrastructure: a NumPy stand-in for one rank's shard, speaking the same protocol as
desc_amd.sharded.HipShard (reset / finish / colsum / sweep / objective / stopped / download),
so that the multi-rank driver and its collectives can be exercised on CPU with gloo.
It restates DESC_PGD.m:185-257 for a contiguous range of edges-with-cycles, using the
oracle's structure arrays.  Never used by the product."""
import numpy as np
import torch

from oracle.desc_pgd_literal import project_simplex_literal


class NumpyShard:
    def __init__(self, st, S0, rank, world):
        self.st, self.d = st, np.asarray(S0)
        self.rank, self.world = rank, world
        mp, m = st["m_pos"], st["m"]
        cyc = st["cum_ind"]
        # contiguous split of the edges-with-cycles, balanced by cycle count
        targets = [cyc[-1] * r // world for r in range(world + 1)]
        self.rank_seg = [int(np.searchsorted(cyc, t, side="left")) for t in targets]
        self.rank_seg[-1] = mp
        self.lo, self.hi = self.rank_seg[rank], self.rank_seg[rank + 1]
        self.c_lo, self.c_hi = int(cyc[self.lo]), int(cyc[self.hi])
        self.slice_len = max(self.rank_seg[r + 1] - self.rank_seg[r] for r in range(world)) + 2
        # exchange layout of the mirror sums: part r = {T1, T2} of the edges rank r owns, padded to the largest shard
        self.t_part = 2 * max(max(self.rank_seg[r + 1] - self.rank_seg[r] for r in range(world)), 1)
        self.T = torch.zeros(world * self.t_part + 1, dtype=torch.float64)      # send
        self.T_recv = torch.zeros(self.t_part, dtype=torch.float64)
        owner = np.searchsorted(np.asarray(self.rank_seg[1:]), np.arange(mp), side="right")
        self.xpos = owner * self.t_part + 2 * (np.arange(mp) - np.asarray(self.rank_seg)[owner])     # position of T1 of every edge
        self.sall = torch.zeros(world * self.slice_len, dtype=torch.float64)
        self.seg = np.repeat(np.arange(mp), np.diff(cyc))
        self.m = m

    # ---- protocol
    def reset(self, params):
        st = self.st
        self.p = params
        self.t = 0
        self.tp = params.t0
        self.S = np.ones(self.m)
        self.w = np.zeros(st["m_cycle"])                             # only [c_lo, c_hi) is meaningful
        cnt = np.diff(st["cum_ind"])
        loc = slice(self.c_lo, self.c_hi)
        self.w[loc] = 1.0 / cnt[self.seg[loc]]
        self.S_loc = np.add.reduceat(self.w[loc] * self.d[loc], st["cum_ind"][self.lo:self.hi] - self.c_lo) if self.hi > self.lo else np.zeros(0)
        self.obj, self.avg = [], []
        self.misses, self.stop, self.iters_run = 0, False, 0
        self.part = (0.0, 0.0)
        self.S_prev = None

    def _pack(self):
        sl = self.sall.numpy()[self.rank * self.slice_len:(self.rank + 1) * self.slice_len]
        sl[:self.hi - self.lo] = self.S_loc
        sl[-2:] = self.part

    def _unpack(self):
        S = np.ones(self.m)
        pairs = []
        for r in range(self.world):
            sl = self.sall.numpy()[r * self.slice_len:(r + 1) * self.slice_len]
            lo, hi = self.rank_seg[r], self.rank_seg[r + 1]
            S[self.st["pos_edge"][lo:hi]] = sl[:hi - lo]
            pairs.append((sl[-2], sl[-1]))
        return S, pairs

    def finish(self, initial=0):
        if initial == 1:
            self._pack(); return
        S, pairs = self._unpack()
        if initial == 2:
            self.S = S; self.S_hist = [S.copy()]; return
        if self.stop:
            return
        self.S_hist.append(S.copy()); self.S = S
        t = self.t
        obj_prev = sum(p[0] for p in pairs); chg = sum(p[1] for p in pairs)
        self.avg.append(chg / self.m)
        self._record(t - 1, obj_prev)

    def _record(self, it, obj):
        if it < 1:
            return
        self.obj.append(obj)
        if it > 1 and self.obj[it - 2] - self.obj[it - 1] < self.p.stop_tol:
            self.misses += 1
            if self.misses >= self.p.patience:
                self.stop, self.iters_run = True, it
        else:
            self.misses = 0

    def colsum(self):
        """partial mirror sums over the mirror cycles this rank owns"""
        if self.stop:
            return
        st = self.st
        T = np.zeros(2 * st["m_pos"])
        for col, key in ((0, "ikj"), (1, "jki")):
            idx = st[key]
            ok = (idx >= self.c_lo) & (idx < self.c_hi)
            T[col * st["m_pos"]:(col + 1) * st["m_pos"]] = np.bincount(self.seg[ok], self.w[idx[ok]], minlength=st["m_pos"])
        out = self.T.numpy(); out[:] = 0.0
        out[self.xpos] = T[:st["m_pos"]]; out[self.xpos + 1] = T[st["m_pos"]:]

    def sweep(self):
        if self.stop:
            return
        st, p = self.st, self.p
        self.t += 1; self.tp += 1
        mp = st["m_pos"]
        T = np.zeros(2 * mp)                                            # totals of the owned edges, from the reduce-scatter
        nl = self.hi - self.lo
        T[self.lo:self.hi] = self.T_recv.numpy()[0:2 * nl:2]; T[mp + self.lo:mp + self.hi] = self.T_recv.numpy()[1:2 * nl:2]
        step = p.lr
        if p.step_kind == 1:
            step = p.lr / (np.trunc(self.tp / p.decay_interval) + 1)
        loc = np.arange(self.c_lo, self.c_hi)
        S = self.S
        ssum = S[st["e_jk"][loc]] + S[st["e_ki"][loc]]
        obj_prev = float(self.w[loc] @ ssum)
        g = ssum + ((st["ikj"][loc] >= 0) * T[self.seg[loc]] + (st["jki"][loc] >= 0) * T[mp + self.seg[loc]]) * self.d[loc]
        new_S = np.zeros(self.hi - self.lo)
        chg = 0.0
        for l in range(self.lo, self.hi):
            a, b = st["cum_ind"][l] - self.c_lo, st["cum_ind"][l + 1] - self.c_lo
            gl = g[a:b]; cnt = b - a
            nv = np.ones(cnt) / cnt ** 0.5
            gl = gl - (gl @ nv) * nv
            wn = project_simplex_literal(self.w[self.c_lo + a:self.c_lo + b] - step * gl)
            self.w[self.c_lo + a:self.c_lo + b] = wn
            new_S[l - self.lo] = wn @ self.d[self.c_lo + a:self.c_lo + b]
            chg += abs(new_S[l - self.lo] - S[st["pos_edge"][l]])
        self.S_loc = new_S
        self.part = (obj_prev, chg)
        self._pack()

    def objective(self, phase):
        if phase == 0:
            st = self.st
            loc = np.arange(self.c_lo, self.c_hi)
            self.part = (float(self.w[loc] @ (self.S[st["e_jk"][loc]] + self.S[st["e_ki"][loc]])), 0.0)
            sl = self.sall.numpy()[self.rank * self.slice_len:(self.rank + 1) * self.slice_len]
            sl[-2:] = self.part
        elif not self.stop:
            _, pairs = self._unpack()
            self._record(self.t, sum(p[0] for p in pairs))

    def stopped(self):
        return self.stop

    def download(self):
        it = self.iters_run if self.stop else self.t
        return dict(S_vec=self.S_hist[it], obj=np.array(self.obj[:it]), avg=np.array(self.avg[:it]), iters_run=it)
