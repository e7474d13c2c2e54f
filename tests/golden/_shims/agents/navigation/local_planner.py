"""Stand-in for CARLA's PythonAPI agents.navigation.local_planner (absent here): only the RoadOption enum that
team_code_v2/waypointer.py:4 imports, with CARLA 0.9.10's values."""
from enum import Enum


class RoadOption(Enum):
    VOID = -1
    LEFT = 1
    RIGHT = 2
    STRAIGHT = 3
    LANEFOLLOW = 4
    CHANGELANELEFT = 5
    CHANGELANERIGHT = 6
