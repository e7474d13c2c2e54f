"""Host-side glue of LAVAgent (lav_amd/agent/) against fixtures produced by the REFERENCE's own ekf.py / pid.py /
planner.py / waypointer.py on a seeded route (tests/golden/make_golden.py:gold_agent).  CPU only."""
import numpy as np

from lav_amd import synth
from lav_amd.agent import EKF, PIDController, RoadOption, RoutePlanner, VehicleControl, Waypointer
from tests.util import crc


def _plan(sc):
    return [({"lat": la, "lon": lo, "z": 0.0}, RoadOption(int(c))) for la, lo, c in zip(sc["lat"], sc["lon"], sc["cmds"])]


def test_scenario_is_the_one_the_fixture_was_made_from(golden):
    assert crc(synth.agent_scenario()["gps"]) == int(golden["agent"]["gps_crc"][0])


def test_ekf_matches_reference(golden):
    sc = synth.agent_scenario()
    ekf = EKF(1, 1.477531, 1.393600)
    ekf.init(*sc["ekf_gps"][0], sc["ekf_compass"][0] - np.pi / 2)
    xs = []
    for (spd, steer), (la, lo), comp in zip(sc["ekf_in"], sc["ekf_gps"], sc["ekf_compass"]):
        ekf.step(spd, steer, la, lo, comp - np.pi / 2)
        xs.append(ekf.x.copy())
    ref = golden["agent"]["ekf_x"]
    np.testing.assert_allclose(np.array(xs), ref, rtol=1e-12, atol=1e-9)
    assert np.abs(np.diff(ref[:, 0])).max() > 0.01          # the state really moves


def test_pid_matches_reference(golden):
    sc = synth.agent_scenario()
    a, b = PIDController(K_P=0.8, K_I=0.5, K_D=0.2, n=40), PIDController(K_P=5.0, K_I=0.5, K_D=1.0, n=3)
    np.testing.assert_allclose([a.step(e) for e in sc["pid_err"]], golden["agent"]["pid_a"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose([b.step(e) for e in sc["pid_err"]], golden["agent"]["pid_b"], rtol=1e-13, atol=1e-13)


def test_route_planner_and_waypointer_match_reference(golden):
    sc = synth.agent_scenario()
    plan = _plan(sc)
    rp = RoutePlanner(plan)
    wp = Waypointer(plan, sc["gps"][0], pop_lane_change=True)
    wp2 = Waypointer(plan, sc["gps"][0], pop_lane_change=False, pop_turning=True)
    r, w, w2 = [], [], []
    for g in sc["gps"]:
        r.append(rp.run_step(g))
        dx, dy, c = wp.tick(g); w.append([dx, dy, c.value])
        dx, dy, c = wp2.tick(g); w2.append([dx, dy, c.value])
    g = golden["agent"]
    np.testing.assert_allclose(np.array(r), g["route"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(w), g["waypointer"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(w2), g["waypointer_turn"], rtol=0, atol=1e-9)
    # the fixture exercises every command and the look-ahead lane-change jump
    assert set(g["waypointer"][:, 2].astype(int)) >= {1, 2, 3, 4, 5, 6}
    assert rp.current_idx > 30 and wp.current_idx > 30


def test_agent_surface_without_gpu():
    """Entry point, sensor suite and the no-device failure mode (the path has no CPU fallback)."""
    import pytest
    import torch
    from lav_amd import lav_agent
    assert lav_agent.get_entry_point() == "LAVAgent"
    agent = lav_agent.LAVAgent.__new__(lav_agent.LAVAgent)
    agent.camera_x, agent.camera_z = 1.5, 2.4
    ids = [s["id"] for s in agent.sensors()]
    assert ids == ["EGO", "GPS", "IMU", "LIDAR", "RGB_0", "RGB_1", "RGB_2", "TEL_RGB"]
    cams = {s["id"]: s for s in agent.sensors() if s["type"] == "sensor.camera.rgb"}
    assert [cams[f"RGB_{i}"]["yaw"] for i in range(3)] == [-60, 0, 60]
    assert (cams["RGB_0"]["width"], cams["RGB_0"]["height"], cams["RGB_0"]["fov"]) == (256, 288, 64)
    assert (cams["TEL_RGB"]["width"], cams["TEL_RGB"]["height"], cams["TEL_RGB"]["fov"]) == (480, 288, 40)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            lav_agent.LAVAgent(None)
    c = VehicleControl(steer=0.1, throttle=0.2, brake=0.0)
    assert (c.steer, c.throttle, c.brake) == (0.1, 0.2, 0.0)
