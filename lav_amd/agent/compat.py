"""The three third-party names the agent needs - carla.VehicleControl, leaderboard's AutonomousAgent/Track and
agents.navigation.local_planner.RoadOption - taken from the real packages when they are importable (inside a CARLA
leaderboard container) and otherwise provided as minimal stand-ins with the same attributes
(team_code_v2/lav_agent_fast.py:8,15; team_code_v2/waypointer.py:4)."""
from __future__ import annotations

import enum
from dataclasses import dataclass

try:  # pragma: no cover - only inside a CARLA install
    from carla import VehicleControl  # type: ignore
except Exception:
    @dataclass
    class VehicleControl:
        throttle: float = 0.0
        steer: float = 0.0
        brake: float = 0.0
        hand_brake: bool = False
        reverse: bool = False
        manual_gear_shift: bool = False
        gear: int = 0

try:  # pragma: no cover
    from leaderboard.autoagents.autonomous_agent import AutonomousAgent, Track  # type: ignore
except Exception:
    class Track(enum.Enum):
        SENSORS = "SENSORS"
        MAP = "MAP"

    class AutonomousAgent:
        """What leaderboard's base class gives an agent: the constructor calls setup(), the evaluator sets the global
        plan and calls the instance once per tick."""

        def __init__(self, path_to_conf_file=None):
            self.track = Track.SENSORS
            self._global_plan = None
            self._global_plan_world_coord = None
            self.sensor_interface = None
            self.setup(path_to_conf_file)

        def setup(self, path_to_conf_file):
            pass

        def sensors(self):
            return []

        def run_step(self, input_data, timestamp):
            return VehicleControl()

        def destroy(self):
            pass

        def set_global_plan(self, global_plan_gps, global_plan_world_coord=None):
            self._global_plan = list(global_plan_gps)
            self._global_plan_world_coord = global_plan_world_coord

try:  # pragma: no cover
    from agents.navigation.local_planner import RoadOption  # type: ignore
except Exception:
    class RoadOption(enum.Enum):
        VOID = -1
        LEFT = 1
        RIGHT = 2
        STRAIGHT = 3
        LANEFOLLOW = 4
        CHANGELANELEFT = 5
        CHANGELANERIGHT = 6
