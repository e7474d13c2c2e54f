"""No-op stand-in for wandb (team_code_v2/lav_agent_fast.py:5,80,163: init / log / Video)."""


def init(*a, **k):
    return None


def log(*a, **k):
    return None


class Video:
    def __init__(self, *a, **k):
        pass
