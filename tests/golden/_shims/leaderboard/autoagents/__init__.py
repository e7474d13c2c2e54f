"""Stand-in for the CARLA leaderboard package (absent here): only autoagents.autonomous_agent, which
team_code_v2/lav_agent_fast.py:15 imports its base class from."""
