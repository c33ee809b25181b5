import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from desc_amd import _lib
from desc_amd.sharded import HipShard
from tests.helpers import make_problem, c_params
from oracle import oracle as O
mo, nn, ii, jj, rij = make_problem("uniform", n=60, p=0.5, q=0.2, sigma=0.1, seed=5)
st = O.build_structure(nn, ii, jj, seed=9); S0 = O.cycle_d(ii, jj, rij.reshape(-1, 9), st)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
def run(world, iters):
    s_ = _lib.Structure.build(prob, 30, 9, _lib.BUILD_HOST, 0)
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    sh = [HipShard(prob, s_, 0, r, world, stream=stream) for r in range(world)]
    L = sh[0].slice_len
    p = c_params(max(iters,1), lr=0.01, seed=9)
    Ts = []
    with torch.cuda.stream(stream):
        for s in sh: s.reset(p)
        for s in sh: s.finish(1)
        for r in range(world):
            piece = sh[r].sall.view(world, L)[r].clone()
            for s in sh: s.sall.view(world, L)[r].copy_(piece)
        for s in sh: s.finish(2)
        for it in range(iters):
            for s in sh: s.colsum()
            tot = torch.zeros_like(sh[0].T)
            for s in sh: tot += s.T
            for s in sh: s.T.copy_(tot)
            Ts.append(tot.cpu().numpy().copy())
            for s in sh: s.sweep()
            for r in range(world):
                piece = sh[r].sall.view(world, L)[r].clone()
                for s in sh: s.sall.view(world, L)[r].copy_(piece)
            for s in sh: s.finish(0)
        for s in sh: s.objective(0)
        for r in range(world):
            piece = sh[r].sall.view(world, L)[r].clone()
            for s in sh: s.sall.view(world, L)[r].copy_(piece)
        for s in sh: s.objective(1)
        outs = [s.download() for s in sh]
    print(world, [ (s.info.seg_lo, s.info.seg_hi, s.info.cyc_lo, s.info.cyc_hi) for s in sh], 'slice', L)
    return outs, Ts
for iters in (0, 1, 2):
    ref = O.pgd_run(st, S0, iters, lr=0.01) if iters else None
    o1, T1 = run(1, iters); o2, T2 = run(2, iters)
    print('iters', iters, 'S w1 vs w2', np.abs(o1[0]['S_vec'] - o2[0]['S_vec']).max(), 'rank1 vs rank0', np.abs(o2[0]['S_vec'] - o2[1]['S_vec']).max(),
          'T diff', [float(np.abs(a - b).max()) for a, b in zip(T1, T2)], ('vs oracle', np.abs(o1[0]['S_vec'] - ref['S_vec']).max(), np.abs(o2[0]['S_vec'] - ref['S_vec']).max()) if ref else '')
