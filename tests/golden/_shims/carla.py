"""Stand-in for the CARLA 0.9.10.1 Python API (absent here).  Only what
team_code_v2/model_inference.py:267-274 and team_code_v2/point_painting.py:14-21
touch: Location, Rotation, Transform.get_matrix / get_inverse_matrix.

Restates LibCarla/source/carla/geom/Transform.h (GetMatrix / GetInverseMatrix /
InverseTransformPoint) in float32, as LibCarla computes them.  Restated from the
published source from memory - no CARLA wheel is available to cross-check
("parity unpinned" for this third-party piece).
"""
import numpy as np

f32 = np.float32


class Location:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = f32(x), f32(y), f32(z)


class Rotation:
    def __init__(self, pitch=0.0, yaw=0.0, roll=0.0):
        self.pitch, self.yaw, self.roll = f32(pitch), f32(yaw), f32(roll)


class Transform:
    def __init__(self, location=None, rotation=None):
        self.location = location if location is not None else Location()
        self.rotation = rotation if rotation is not None else Rotation()

    def _cs(self):
        to_rad = f32(np.pi) / f32(180.0)
        r = self.rotation
        cy, sy = np.cos(r.yaw * to_rad, dtype=f32), np.sin(r.yaw * to_rad, dtype=f32)
        cr, sr = np.cos(r.roll * to_rad, dtype=f32), np.sin(r.roll * to_rad, dtype=f32)
        cp, sp = np.cos(r.pitch * to_rad, dtype=f32), np.sin(r.pitch * to_rad, dtype=f32)
        return cy, sy, cr, sr, cp, sp

    def get_matrix(self):
        cy, sy, cr, sr, cp, sp = self._cs()
        l = self.location
        m = [[cp * cy, cy * sp * sr - sy * cr, -cy * sp * cr - sy * sr, l.x],
             [cp * sy, sy * sp * sr + cy * cr, -sy * sp * cr + cy * sr, l.y],
             [sp, -cp * sr, cp * cr, l.z],
             [f32(0), f32(0), f32(0), f32(1)]]
        return [[float(v) for v in row] for row in m]

    def get_inverse_matrix(self):
        cy, sy, cr, sr, cp, sp = self._cs()
        l = self.location
        ax, ay, az = -l.x, -l.y, -l.z
        r0 = (cp * cy, cp * sy, sp)
        r1 = (cy * sp * sr - sy * cr, sy * sp * sr + cy * cr, -cp * sr)
        r2 = (-cy * sp * cr - sy * sr, -sy * sp * cr + cy * sr, cp * cr)
        tx = ax * r0[0] + ay * r0[1] + az * r0[2]
        ty = ax * r1[0] + ay * r1[1] + az * r1[2]
        tz = ax * r2[0] + ay * r2[1] + az * r2[2]
        m = [[r0[0], r0[1], r0[2], tx],
             [r1[0], r1[1], r1[2], ty],
             [r2[0], r2[1], r2[2], tz],
             [f32(0), f32(0), f32(0), f32(1)]]
        return [[float(v) for v in row] for row in m]


class VehicleControl:
    """carla.VehicleControl: the return type of run_step (lav_agent_fast.py:232,360)."""

    def __init__(self, throttle=0.0, steer=0.0, brake=0.0, hand_brake=False, reverse=False, manual_gear_shift=False, gear=0):
        self.throttle, self.steer, self.brake = float(throttle), float(steer), float(brake)
        self.hand_brake, self.reverse, self.manual_gear_shift, self.gear = hand_brake, reverse, manual_gear_shift, gear
