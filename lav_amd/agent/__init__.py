"""Host-side (CPU) glue of the v2 leaderboard agent: pose filter, controllers, route following, the compat layer for
the CARLA / leaderboard packages.  None of this is on the GPU hot path; it exists so that lav_amd.lav_agent.LAVAgent is
a drop-in for team_code_v2/lav_agent_fast.py (level B1)."""
from .compat import AutonomousAgent, RoadOption, Track, VehicleControl  # noqa: F401
from .ekf import EKF  # noqa: F401
from .pid import PIDController  # noqa: F401
from .route import RoutePlanner, Waypointer  # noqa: F401
