"""Stand-in for OpenCV: team_code_v2/lav_agent_fast.py only draws its debug video with it (visualize, :459-517).
The fixture generator replaces visualize(); the one name evaluated at class-definition time is the font constant."""
FONT_HERSHEY_SIMPLEX = 0
COLOR_GRAY2RGB = 8
