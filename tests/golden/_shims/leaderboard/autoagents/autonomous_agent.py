"""What leaderboard/autoagents/autonomous_agent.py gives an agent (CARLA leaderboard, branch `stable`; restated from
its published interface): the constructor calls setup(), the evaluator hands over the global plan, `__call__` wraps
run_step.  Only what team_code_v2/lav_agent_fast.py touches."""
import enum


class Track(enum.Enum):
    SENSORS = "SENSORS"
    MAP = "MAP"


class AutonomousAgent:
    def __init__(self, path_to_conf_file):
        self.track = Track.SENSORS
        self._global_plan = None
        self._global_plan_world_coord = None
        self.sensor_interface = None
        self.setup(path_to_conf_file)

    def setup(self, path_to_conf_file):
        pass

    def set_global_plan(self, global_plan_gps, global_plan_world_coord=None):
        self._global_plan = list(global_plan_gps)
        self._global_plan_world_coord = global_plan_world_coord
