"""Pose filter of the agent: kinematic-bicycle prediction + GNSS/compass correction (team_code_v2/ekf.py:4-107).

State (x, y, theta) in the local metric frame x = R*lat, y = R*lon*cos(cos0) (R = 6371 km; the reference passes
cos0 = 1, i.e. the literal cos(1 rad) - kept).  F = H = I: the covariance recursion does not linearise the motion
model, exactly like the reference."""
from __future__ import annotations

import math

import numpy as np

EARTH_RADIUS = 6371e3


class EKF:
    def __init__(self, cos0, lf, lr, gnss_noise=0.000005, compass_noise=1e-7, max_steer_angle=70, freq=20):
        sigma_xy = EARTH_RADIUS * gnss_noise * math.pi / 180.0      # metres
        sigma_th = compass_noise * math.pi / 180.0                  # radians
        self.Q = 1e-7 * np.eye(3)
        self.R = np.diag([sigma_xy ** 2, sigma_xy ** 2, sigma_th ** 2])
        self.x = np.zeros(3)
        self.P = np.zeros((3, 3))
        self.max_steer_angle = math.radians(max_steer_angle)
        self.cos0, self.lr, self.L = cos0, lr, lf + lr
        self.dt = 1.0 / freq

    def latlon_to_xy(self, lat, lon):
        # same association as the reference: (R * lat) * (pi / 180)
        return EARTH_RADIUS * lat * (math.pi / 180), EARTH_RADIUS * lon * (math.pi / 180) * math.cos(self.cos0)

    def init(self, lat, lon, compass):
        self.x[:2] = self.latlon_to_xy(lat, lon)
        self.x[2] = compass
        self.P = np.zeros((3, 3))

    def kbm_step(self, spd, steer):
        """One tick of the kinematic bicycle model.  The yaw-rate term uses tan(theta) (sic, ekf.py:90)."""
        px, py, th = self.x
        beta = np.arctan(self.lr * np.tan(steer * self.max_steer_angle) / self.L)
        return np.array([px + spd * math.cos(th + beta) * self.dt,
                         py + spd * math.sin(th + beta) * self.dt,
                         th + spd * np.tan(th) * np.cos(beta) / self.L * self.dt])

    def step(self, spd, steer, lat, lon, compass):
        z = np.array([*self.latlon_to_xy(lat, lon), compass])
        x_pred = self.kbm_step(spd, steer)
        P_pred = self.P + self.Q                                    # F = I
        K = P_pred @ np.linalg.inv(P_pred + self.R)                 # H = I
        self.x = x_pred + K @ (z - x_pred)
        self.P = (np.eye(3) - K) @ P_pred
