import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from desc_amd import _lib
from tests.helpers import make_problem, c_params
mo, nn, ii, jj, rij = make_problem("uniform", n=300, p=0.5, seed=1)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
def once(where):
    st = _lib.Structure.build(prob, 30, 0, where, 0)
    s = _lib.Solver(prob, st, 0)
    out = s.run(c_params(5))
    if where == _lib.BUILD_DEVICE: st.arrays()
    s.destroy(); st.free()
    _lib.spectral_run(prob); _lib.cemp_run(prob, [1, 2], 2, 20)
for w in (_lib.BUILD_DEVICE, _lib.BUILD_HOST): once(w)
torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
for i in range(60):
    once(_lib.BUILD_DEVICE if i % 2 == 0 else _lib.BUILD_HOST)
torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB after %.1f MB delta %.2f MB" % (f0 / 1e6, f1 / 1e6, (f0 - f1) / 1e6))
