"""Hand-derived known answer for the sampling regime of DESC_PGD.m:185-230 (mirror cycles absent).

K4 with rotations about the z axis, so every cycle inconsistency is a decimal fraction chosen by hand:
theta_12 = 0.2 pi, theta_23 = 0.1 pi, theta_24 = 0.3 pi, all other edges the identity, hence
    d(123) = 0.3   d(124) = 0.5   d(134) = 0   d(234) = 0.2
Two cycles are removed from the sample -- (13;4) and (24;1) -- which is what `datasample`
(DESC_PGD.m:84) does to an edge on a large graph; the reference itself never samples an edge with
fewer than 30 common neighbours, so the structure is handed over (literal oracle: `forced_lists`;
C oracle and HIP library: an imported structure).  Cycle numbering (0-based), segment by segment:
    E12: c0=(12;3) c1=(12;4) | E13: c2=(13;2) | E14: c3=(14;2) c4=(14;3)
    E23: c5=(23;1) c6=(23;4) | E24: c7=(24;3) | E34: c8=(34;1) c9=(34;2)
Mirror maps (DESC_PGD.m:103-127), X = not sampled:
    c   IKJ  JKI        c   IKJ  JKI
    c0  c2   c5         c5  c0   c2
    c1  c3   X          c6  c7   c9
    c2  c0   c5         c7  c6   c9
    c3  c1   X          c8  X    c4
    c4  X    c8         c9  c6   c7
The tables below were worked out with pencil-and-paper decimal arithmetic from the text of
DESC_PGD.m:148-157 (initial state) and :185-233 (two iterations, ConstantStepSize(2)); nothing in
this file calls a restatement.  The step is large on purpose: the simplex projection (:215-224)
clips in iteration 1 (segment E23) and in iteration 2 (E14, E23, E34).

Iteration 1, from w = (.5 .5 | 1 | .5 .5 | .5 .5 | 1 | .5 .5), S = (.4 .3 .25 .25 .2 .1):
    edge  T1 = sum w(IKJ(mask))   T2 = sum w(JKI(mask))
    E12   w2+w3 = 1.5             w5 = .5        (c1 has no JKI: T2 is not added to c1)
    E13   w0 = .5                 w5 = .5
    E14   w1 = .5 (c3 only)       w8 = .5 (c4 only)
    E23   w0+w7 = 1.5             w2+w9 = 1.5
    E24   w6 = .5                 w9 = .5
    E34   w6 = .5 (c9 only)       w4+w7 = 1.5
    g (:193) = 1.15 1.2 | .95 | .85 .4 | 1.6 .9 | .55 | .55 .85
    after mean removal (:199-203) = -.025 .025 | 0 | .225 -.225 | .35 -.35 | 0 | -.15 .15
    w - 2g = .55 .45 | 1 | .05 .95 | -.2 1.2 | 1 | .8 .2 ; E23 projects to (0, 1) with T = .2
Iteration 2:
    T1, T2 = E12 1.05, 0 | E13 .55, 0 | E14 .45, .8 | E23 1.55, 1.2 | E24 1, .2 | E34 1, 1.95
    g = .815 .75 | .755 | .815 .34 | 1.515 .79 | .48 | .325 .99
    w - 2g = .485 .515 | 1 | -.425 1.425 | -.725 1.725 | 1 | 1.465 -.465
"""
import numpy as np

EDGES = [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]
THETA = {(1, 2): 0.2, (2, 3): 0.1, (2, 4): 0.3}          # in units of pi; the other edges are the identity
FORCED = {2: [2], 5: [3]}                                 # 1-based edge index -> kept third vertices
LR = 2.0

# 0-based imported structure
POS_EDGE = np.arange(6, dtype=np.int32)
CUM_IND = np.array([0, 2, 3, 5, 7, 8, 10], dtype=np.int64)
K = np.array([3, 4, 2, 2, 3, 1, 4, 3, 1, 2], dtype=np.int32) - 1
E_JK = np.array([3, 4, 3, 4, 5, 1, 5, 5, 2, 4], dtype=np.int32)      # Ind_jk (edge ids 0..5 = E12 E13 E14 E23 E24 E34)
E_KI = np.array([1, 2, 0, 0, 1, 0, 4, 3, 1, 3], dtype=np.int32)      # Ind_ki
IKJ = np.array([2, 3, 0, 1, -1, 0, 7, 6, -1, 6], dtype=np.int32)
JKI = np.array([5, -1, 5, -1, 8, 2, 9, 9, 4, 7], dtype=np.int32)

D = np.array([.3, .5, .3, .5, 0, .3, .2, .2, 0, .2])                  # S0_long
W0 = np.array([.5, .5, 1, .5, .5, .5, .5, 1, .5, .5])
S_INIT = np.array([.4, .3, .25, .25, .2, .1])
W1 = np.array([.55, .45, 1, .05, .95, 0, 1, 1, .8, .2])
S1 = np.array([.39, .3, .025, .2, .2, .04])
OBJ1, AVG1 = 2.13875, 0.0575
W2 = np.array([.485, .515, 1, 0, 1, 0, 1, 1, 1, 0])
S2 = np.array([.403, .3, 0, .2, .2, 0])
OBJ2, AVG2 = 1.9485, 0.013
TOL = 1e-12           # the only inexact step is acos(cos(x)) for the three non-trivial angles

# First GetStep of HybridGradient (Utils/HybridGradient.m:23-41, strategy 0 = Adam) on the same structure: with m_0 = v_0 = 0,
#   m_1 = (1-b1) g, v_1 = (1-b2) g.^2, m_1/(1-b1^1) = g, v_1/(1-b2^1) = g.^2  =>  step = -lr * g ./ (|g| + 1e-8),
# i.e. every cycle moves by lr against the sign of its (mean-removed) gradient, whatever beta_1, beta_2.  From the table above
# g = -.025 .025 | 0 | .225 -.225 | .35 -.35 | 0 | -.15 .15 ; with lr = 0.1 nothing leaves the simplex:
ADAM_LR = 0.1
_G1 = np.array([-.025, .025, 0, .225, -.225, .35, -.35, 0, -.15, .15])
W1_ADAM = W0 - ADAM_LR * _G1 / (np.abs(_G1) + 1e-8)        # = .6 .4 | 1 | .4 .6 | .4 .6 | 1 | .6 .4  up to 1e-8 / |g| relative
S1_ADAM = np.array([W1_ADAM[0] * .3 + W1_ADAM[1] * .5, .3, W1_ADAM[3] * .5, W1_ADAM[5] * .3 + W1_ADAM[6] * .2, .2, W1_ADAM[9] * .2])
ADAM_M1 = 0.1 * _G1                                          # (1 - beta_1) g with beta_1 = 0.9
ADAM_V1 = 0.001 * _G1 ** 2                                   # (1 - beta_2) g.^2 with beta_2 = 0.999
assert np.allclose(W1_ADAM, [.6, .4, 1, .4, .6, .4, .6, 1, .6, .4], atol=1e-6) and np.allclose(S1_ADAM, [.38, .3, .2, .24, .2, .08], atol=1e-6)


def rotations():
    """RijMat (3,3,6): rotation about z by theta_ij."""
    R = np.zeros((3, 3, 6))
    for l, e in enumerate(EDGES):
        t = THETA.get(e, 0.0) * np.pi
        R[:, :, l] = [[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]] if t else np.eye(3)
    return R


def structure_dict():
    return dict(n=4, m=6, m_pos=6, m_cycle=10, n_sample=30, pos_edge=POS_EDGE, cum_ind=CUM_IND, k=K, e_jk=E_JK, e_ki=E_KI,
                ikj=IKJ, jki=JKI)
