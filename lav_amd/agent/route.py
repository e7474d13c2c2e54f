"""Route following on the GNSS global plan: the next-waypoint tracker (team_code_v2/planner.py:7-58) and the
high-level command tracker (team_code_v2/waypointer.py:7-103).  Both work in the local metric frame
x = R*lat, y = R*lon*cos(mean latitude of the plan)."""
from __future__ import annotations

import math

import numpy as np

from .compat import RoadOption

EARTH_RADIUS = 6371e3


class _PlanFrame:
    def __init__(self, global_plan):
        self.cos_0 = sum(g["lat"] * (math.pi / 180) for g, _ in global_plan) / len(global_plan)

    def latlon_to_xy(self, lat, lon):
        # same association as the reference: (R * lat) * (pi / 180)
        return EARTH_RADIUS * lat * (math.pi / 180), EARTH_RADIUS * lon * (math.pi / 180) * math.cos(self.cos_0)


class RoutePlanner(_PlanFrame):
    """run_step(gnss) -> vector to the current checkpoint; the checkpoint advances to the NEXT route node (only) once
    the vehicle is within curr_threshold of the current one and next_threshold of the next."""

    def __init__(self, global_plan, curr_threshold=20, next_threshold=75, debug=False):
        super().__init__(global_plan)
        self.route = [self.latlon_to_xy(g["lat"], g["lon"]) for g, _ in global_plan]
        self.curr_threshold, self.next_threshold = curr_threshold, next_threshold
        self.current_idx = 0
        self.checkpoint = self.route[0]

    def run_step(self, gnss):
        x, y = self.latlon_to_xy(gnss[0], gnss[1])
        here = math.hypot(self.checkpoint[0] - x, self.checkpoint[1] - y)
        nxt = self.current_idx + 1
        if nxt < len(self.route) and here < self.curr_threshold:
            wx, wy = self.route[nxt]
            if math.hypot(wx - x, wy - y) < self.next_threshold:
                self.checkpoint = [wx, wy]
                self.current_idx = nxt
        return np.array(self.checkpoint) - [x, y]


class Waypointer(_PlanFrame):
    """tick(gnss) -> (dx, dy, command) of the active plan node; optionally jumps ahead to an upcoming lane change."""

    def __init__(self, global_plan, current_gnss, threshold_lane=10., threshold_before=4.5, threshold_after=3.0,
                 threshold_max=50., pop_lane_change=True, pop_turning=False):
        super().__init__(global_plan)
        self._before, self._after, self._max = threshold_before, threshold_after, threshold_max
        self._pop_lane_change, self._pop_turning = pop_lane_change, pop_turning
        self.global_plan = [(*self.latlon_to_xy(g["lat"], g["lon"]), cmd) for g, cmd in global_plan]
        cx, cy = self.latlon_to_xy(current_gnss[0], current_gnss[1])
        self.checkpoint = (cx, cy, RoadOption.LANEFOLLOW)
        self.current_idx = -1

    def tick(self, gnss):
        cur_x, cur_y = self.latlon_to_xy(gnss[0], gnss[1])
        far = math.hypot(self.checkpoint[0] - cur_x, self.checkpoint[1] - cur_y) > self._max
        i = len(self.global_plan) - 1          # where the reference's scan ends when nothing fires
        for j, (wx, wy, cmd) in enumerate(self.global_plan):
            entering = self.checkpoint[2] == RoadOption.LANEFOLLOW and cmd != RoadOption.LANEFOLLOW
            near = math.hypot(cur_x - wx, cur_y - wy) < (self._before if entering else self._after)
            if near and j - self.current_idx == 1:
                self.checkpoint, self.current_idx, i = (wx, wy, cmd), self.current_idx + 1, j
                break
            if (self._pop_turning and far and near and j > self.current_idx
                    and cmd in (RoadOption.LEFT, RoadOption.RIGHT)):
                self.checkpoint, self.current_idx, i = (wx, wy, cmd), j, j
                break
        if self._pop_lane_change:
            cmd = self.checkpoint[2]
            for _ in range(3):
                if i + 1 >= len(self.global_plan) or cmd != RoadOption.LANEFOLLOW:
                    break
                wx, wy, wcmd = self.global_plan[i + 1]
                if wcmd in (RoadOption.CHANGELANELEFT, RoadOption.CHANGELANERIGHT):
                    self.checkpoint, self.current_idx = (wx, wy, wcmd), i + 1
                    break
                cmd = wcmd
                i += 1
        wx, wy, cmd = self.checkpoint
        return wx - cur_x, wy - cur_y, cmd
