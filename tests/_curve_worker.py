"""Worker of tests/test_gpu_train.py::test_train_lidar_loss_curve_vs_reference_trainer: the 500 optimisation steps of the
loss-curve comparison in a process of their own with an EMPTY MIOpen user database.  Which convolution algorithms MIOpen picks
depends on what ran before in the process and on the find results on disk, so inside a test session the curve was a function of
the session (round 6: twelve new convolution tests in front of it moved the orientation term from inside its bar to 9 % beyond
it); here it is a function of the code alone.

Second half of the same fix (round 6, second session): an empty database was not enough - with `cudnn.benchmark` off torch still lets
MIOpen TIME the applicable solvers and keep the fastest, so the choice was a measurement and the curve a random sample per process (seven
runs of one build ended at 5.4 ... 12.2 against the reference's 8.7; one in three failed a bar).  `set_deterministic` now also switches
MIOpen to immediate mode (`torch.backends.miopen.immediate`: the solver comes from MIOpen's database / heuristic); four processes of twelve
steps then agree in every bit of every loss term (tools/determinism_xproc.py) and two runs of this worker give the same 500 rows.

    python tests/_curve_worker.py OUT.npz KEY1,KEY2,...
"""
import os
import sys
import tempfile

_db = tempfile.mkdtemp(prefix="lav_curve_miopen_")
os.environ["MIOPEN_USER_DB_PATH"] = _db
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = _db

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np   # noqa: E402
import torch         # noqa: E402


def main():
    from lav_amd.train import LAV, TrainConfig
    from lav_amd.train.run import set_deterministic
    from lav_amd.train.synthetic import synthetic_lidar_batch
    out, keys = sys.argv[1], sys.argv[2].split(",")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    dev = torch.device("cuda")
    set_deterministic(True)          # the curve is then one fixed curve, not a sample of a distribution
    torch.manual_seed(0)
    lav = LAV(TrainConfig(log_inference=False), dev, what="lidar")
    batches = [synthetic_lidar_batch(2, seed=40 + i, max_points=20000, num_objs=3) for i in range(4)]
    rows = []
    for step in range(steps):
        torch.manual_seed(1000 + step)
        info = lav.train_lidar(*batches[step % 4])
        rows.append([info[k] for k in keys])
    np.savez(out, rows=np.array(rows))


if __name__ == "__main__":
    try:
        main()
    finally:
        import shutil
        shutil.rmtree(_db, ignore_errors=True)
