"""Step-size plugins: host-side mirrors of the reference's handle classes
``Utils/ConstantStepSize.m``, ``Utils/PiecewiseStepSize.m`` and
``Utils/HybridGradient.m`` (the value of ``params.Gradient``, DESC_PGD.m:207).

Inside ``DESC_PGD`` the step is applied by the HIP sweep kernel; the objects carry
the plugin's parameters in and its state (``t``, ``m_t``, ``v_t``) out, as MATLAB
handle objects do.  ``GetStep`` is kept so the objects remain usable stand-alone.
"""
from __future__ import annotations

import math

import numpy as np


class ConstantStepSize:
    """ConstantStepSize.m:6-11."""

    def __init__(self, learning_rate):
        self.learning_rate = float(learning_rate)

    def GetStep(self, grad):
        return -self.learning_rate * np.asarray(grad)


class PiecewiseStepSize:
    """PiecewiseStepSize.m:8-18; ``t`` counts GetStep calls over the object's lifetime."""

    def __init__(self, learning_rate, decay_interval):
        self.learning_rate = float(learning_rate)
        self.decay_interval = decay_interval
        self.t = 0

    def GetStep(self, grad):
        self.t += 1
        step_size = self.learning_rate / (math.trunc(self.t / self.decay_interval) + 1)
        return -step_size * np.asarray(grad)


class HybridGradient:
    """HybridGradient.m:13-52: Adam (strategy 0) or 100*lr/(fix(t/decay)+1) (strategy 1)."""

    def __init__(self, lr, beta_1, beta_2, decay_interval):
        self.lr = float(lr)
        self.beta_1 = float(beta_1)
        self.beta_2 = float(beta_2)
        self.decay_interval = decay_interval
        self.t = 0
        self.strategy = 0
        self.m_t = None
        self.v_t = None

    def GetStep(self, grad):
        grad = np.asarray(grad, dtype=np.float64)
        if self.t == 0:
            self.m_t = np.zeros_like(grad)
            self.v_t = np.zeros_like(grad)
        step = None
        if self.strategy == 0:
            self.t += 1
            self.m_t = (self.beta_1 * self.m_t) + (1 - self.beta_1) * grad
            self.v_t = (self.beta_2 * self.v_t) + (1 - self.beta_2) * (grad ** 2)
            corr_m_t = self.m_t / (1 - self.beta_1 ** self.t)
            corr_v_t = self.v_t / (1 - self.beta_2 ** self.t)
            step = -self.lr * corr_m_t / (np.sqrt(corr_v_t) + 10 ** (-8))
        if self.strategy == 1:
            self.t += 1
            step_size = 100 * (self.lr / (math.trunc(self.t / self.decay_interval) + 1))
            step = -step_size * grad
        return step

    def stopAdam(self):
        self.strategy = 1
        return self
