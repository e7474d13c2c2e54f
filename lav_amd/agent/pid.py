"""Windowed PID of the steering / speed loops (team_code_v2/pid.py:4-26): the "integral" is the MEAN of the last n
errors (window pre-filled with zeros) and the derivative the last difference."""
from __future__ import annotations

from collections import deque

import numpy as np


class PIDController:
    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        self.kp, self.ki, self.kd = K_P, K_I, K_D
        self.errors = deque([0] * n, maxlen=n)

    def step(self, error):
        self.errors.append(error)
        mean = float(np.mean(self.errors)) if len(self.errors) > 1 else 0.0
        diff = self.errors[-1] - self.errors[-2] if len(self.errors) > 1 else 0.0
        return self.kp * error + self.ki * mean + self.kd * diff
