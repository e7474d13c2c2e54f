"""Data side of the trainers (CPU): the pure-Python LMDB reader, the OpenCV stand-ins and the loaders of lav_amd.data
against samples drawn from the REFERENCE's own loader classes (tests/golden/datasets.npz, make_golden.py:gold_datasets)."""
import os
import random
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

from lav_amd.data import datasets, image, lmdb_ro
from tests.util import DATASET_CASES, GOLD, dataset_fixture_config

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------------- LMDB
@pytest.mark.parametrize("n", [0, 1, 7, 300, 20000])
def test_lmdb_reader_finds_every_key_of_a_written_environment(tmp_path, n):
    """Tree depths 0-3, values inline and on overflow pages (the loaders' LiDAR sweeps are 40 KB+ records), keys of mixed
    length: every key is found with its bytes, absent keys are not, iteration is in key order."""
    rng = random.Random(n)
    items = {}
    for i in range(n):
        k = f"lidar_{i:05d}".encode() if i % 3 else (f"k{i}".encode() * rng.randint(1, 5))
        size = rng.choice([0, 1, 7, 100, 2000, 2029, 2030, 2031, 5000, 70000]) if i < 40 else rng.choice([3, 20, 50])
        items[k] = bytes(rng.getrandbits(8) for _ in range(size))
    lmdb_ro.write(str(tmp_path / "env"), items.items())
    env = lmdb_ro.open(str(tmp_path / "env"), max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)
    txn = env.begin(write=False)
    assert env.stat()["entries"] == n and env.stat()["depth"] == (0 if n == 0 else 1 if n < 10 else 2 if n < 1000 else 3)
    for k, v in items.items():
        assert txn.get(k) == v
    for k in (b"nope", b"", b"\x00", b"zzzz", b"lidar_0000", b"lidar_00001x"):
        if k not in items:
            assert txn.get(k) is None
    assert list(txn.items()) == sorted(items.items())
    env.close()


def test_lmdb_reader_rejects_what_it_does_not_understand(tmp_path):
    lmdb_ro.write(str(tmp_path / "env"), {b"a": b"1"}.items())
    with pytest.raises(lmdb_ro.Error, match="read-only"):
        lmdb_ro.open(str(tmp_path / "env"), readonly=False)
    with pytest.raises(lmdb_ro.Error, match="read-only"):
        lmdb_ro.open(str(tmp_path / "env")).begin(write=True)
    raw = bytearray(open(tmp_path / "env" / "data.mdb", "rb").read())
    bad = tmp_path / "bad"
    bad.mkdir()
    struct.pack_into("<I", raw, 16, 0x12345678)                 # magic of meta page 0
    open(bad / "data.mdb", "wb").write(raw)
    with pytest.raises(lmdb_ro.Error, match="magic"):
        lmdb_ro.open(str(bad))
    with pytest.raises(lmdb_ro.Error, match="duplicate"):
        lmdb_ro.write(str(tmp_path / "dup"), [(b"k", b"1"), (b"k", b"2")])


def test_lmdb_picks_the_newer_meta_page(tmp_path):
    """A committed transaction alternates between the two meta pages: the reader must follow the higher transaction id."""
    lmdb_ro.write(str(tmp_path / "env"), {b"key": b"value"}.items())
    raw = bytearray(open(tmp_path / "env" / "data.mdb", "rb").read())
    psize = 4096
    meta0, meta1 = bytes(raw[:psize]), bytes(raw[psize:2 * psize])
    swapped = bytearray(raw)
    swapped[:psize], swapped[psize:2 * psize] = meta1, meta0       # live meta (txn 1) now in slot 0
    os.makedirs(tmp_path / "swapped")
    open(tmp_path / "swapped" / "data.mdb", "wb").write(swapped)
    assert lmdb_ro.open(str(tmp_path / "swapped")).begin().get(b"key") == b"value"


def test_lmdb_reader_fuzz(tmp_path):
    """Random key / value populations (hypothesis): whatever the writer lays out, the reader returns it - including keys that are
    prefixes of one another, the 511-byte maximum, empty values and values straddling the inline / overflow threshold."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    counter = [0]

    @settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.dictionaries(st.binary(min_size=1, max_size=511), st.one_of(st.binary(max_size=64), st.binary(min_size=2000, max_size=2100),
                                                                      st.binary(min_size=4000, max_size=9000)), max_size=120),
           st.sampled_from([512, 4096, 16384]))
    def run(items, psize):
        counter[0] += 1
        path = str(tmp_path / f"env{counter[0]}")
        items = {k: v for k, v in items.items() if 8 + len(k) + 8 <= psize // 4}      # a key must leave room for two nodes on a page
        lmdb_ro.write(path, items.items(), psize=psize)
        env = lmdb_ro.open(path, readonly=True, lock=False)
        txn = env.begin()
        assert env.stat()["psize"] == psize and env.stat()["entries"] == len(items)
        for k, v in items.items():
            assert txn.get(k) == v
            assert txn.get(k + b"\x00") == items.get(k + b"\x00") and txn.get(k[:-1]) == items.get(k[:-1])
        assert list(txn.items()) == sorted(items.items())
        env.close()
    run()


def test_synthetic_route_command_line(tmp_path):
    out = subprocess.run([sys.executable, "-m", "lav_amd.data.synthetic_route", str(tmp_path / "d"), "--routes", "1", "--frames", "22", "--points", "300"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr[-1500:]
    txn = lmdb_ro.open(str(tmp_path / "d" / "route_000")).begin()
    assert int(txn.get(b"len")) == 22 and txn.get(b"town") == b"Town01" and len(txn.get(b"lidar_00021")) == 300 * 16
    assert image.imdecode(np.frombuffer(txn.get(b"map_0_00000"), np.uint8), image.IMREAD_GRAYSCALE).shape == (320, 320)


# ------------------------------------------------------------------------------------------------------------ images
def test_warp_affine_identity_quarter_turn_and_small_angles():
    r = np.random.default_rng(0)
    img = (r.uniform(0, 255, (320, 320, 3))).astype(np.uint8)
    assert np.array_equal(image.warp_affine_linear(img, image.rotation_matrix_2d((160, 280), 0.0)), img)
    # 90 degrees counter-clockwise about (160, 280): pixel (x, y) lands on (160 + (y - 280), 280 - (x - 160)), exactly
    out = image.warp_affine_linear(img, image.rotation_matrix_2d((160, 280), 90.0))
    ys, xs = np.mgrid[0:320, 0:320]
    nx, ny = 160 + (ys - 280), 280 - (xs - 160)
    ok = (nx >= 0) & (nx < 320) & (ny >= 0) & (ny < 320)
    assert np.array_equal(out[ny[ok], nx[ok]], img[ys[ok], xs[ok]])
    # a smooth image under a small rotation: within one grey level of exact (float64) bilinear sampling away from the border
    yy, xx = np.mgrid[0:320, 0:320].astype(np.float64)
    smooth = (127 + 100 * np.sin(xx / 23.0) * np.cos(yy / 31.0))
    M = image.rotation_matrix_2d((160, 280), 11.7)
    got = image.warp_affine_linear(smooth.astype(np.uint8), M).astype(np.float64)
    A = np.vstack([M, [0, 0, 1]])
    inv = np.linalg.inv(A)
    sx = inv[0, 0] * xx + inv[0, 1] * yy + inv[0, 2]
    sy = inv[1, 0] * xx + inv[1, 1] * yy + inv[1, 2]
    inside = (sx > 1) & (sx < 318) & (sy > 1) & (sy < 318)
    x0, y0 = np.floor(sx).astype(int).clip(0, 318), np.floor(sy).astype(int).clip(0, 318)
    fx, fy = sx - x0, sy - y0
    s8 = smooth.astype(np.uint8).astype(np.float64)
    exact = (s8[y0, x0] * (1 - fx) * (1 - fy) + s8[y0, x0 + 1] * fx * (1 - fy) + s8[y0 + 1, x0] * (1 - fx) * fy + s8[y0 + 1, x0 + 1] * fx * fy)
    assert np.abs(got - exact)[inside].max() <= 1.0
    assert (got[~((sx > -1) & (sx < 320) & (sy > -1) & (sy < 320))] == 0).all()        # constant border


def test_png_round_trip():
    img = (np.arange(320 * 320).reshape(320, 320) % 251).astype(np.uint8)
    assert np.array_equal(image.imdecode(np.frombuffer(image.imencode_png(img), np.uint8), image.IMREAD_GRAYSCALE), img)


# ----------------------------------------------------------------------------------------------------------- loaders
@pytest.fixture(scope="module")
def routes(tmp_path_factory):
    return dataset_fixture_config(str(tmp_path_factory.mktemp("routes")))


@pytest.mark.parametrize("name,picks", DATASET_CASES)
def test_loaders_return_the_reference_loaders_samples(routes, name, picks):
    """Same synthetic routes, same seeds: every element of every sample equals what the reference's class returned -
    dtype, shape and bits (the loaders do float64 geometry on float32 records: there is nothing to round differently)."""
    gold = np.load(os.path.join(GOLD, "datasets.npz"))
    ds = datasets.LOADERS[name](routes)
    assert len(ds) == int(gold[f"{name}/len"]) == 20
    where = {(os.path.basename(ds.dir_map[i]), ds.idx_map[i]): i for i in range(len(ds))}
    for p in picks:
        route, frame = ("route_000", p) if p < 10 else ("route_001", p - 10)
        torch.manual_seed(1000 + frame)
        np.random.seed(1000 + frame)
        sample = ds[where[(route, frame)]]
        assert len(sample) == (9 if "bev" in name else 14)
        for k, v in enumerate(sample):
            v = np.asarray(v)
            if name in ("lidar", "lidar_painted") and k == 0:
                v = v[:sample[1]]
            want = gold[f"{name}/{route}/{frame}/{k}"]
            assert v.dtype == want.dtype and v.shape == want.shape, (name, p, k, v.dtype, want.dtype, v.shape, want.shape)
            assert np.array_equal(v, want), f"{name} sample {p} element {k}: max |diff| {np.abs(v.astype(np.float64) - want.astype(np.float64)).max()}"


def test_samples_look_like_driving_data(routes):
    """Sanity of what the fixture exercises: actors that leave mid-route are filtered, sweeps are stacked with a one-hot time
    channel, the painted scores are masked to the cameras' field of view, targets land inside the maps."""
    ds = datasets.TemporalLiDARPaintedDataset(routes)
    torch.manual_seed(5)
    np.random.seed(5)
    lidar, n, heat, size, ori, bev, ego_locs, cmd, nxp, bra, locs, oris, typs, n_obj = ds[12]
    assert n == 6000 and lidar.shape == (6000, 11) and set(np.unique(lidar[:, 8:].sum(1))) == {1.0}
    assert (lidar[:, 8:].sum(0) > 0).all()                                   # all three sweeps survive the shuffle + cut
    behind = lidar[:, 0] < -3                                                  # points behind the car are in no camera
    assert behind.any() and (lidar[behind][:, 4:8] == 0).all() and (lidar[~behind][:, 4:8] > 0).any()
    assert 2 <= n_obj <= 8 and float(heat.max()) == pytest.approx(1.0, abs=0.3) and bev.shape == (9, 320, 320) and bev[0].any() and bev[3].any()
    assert np.allclose(ego_locs[0], 0) and np.linalg.norm(ego_locs[-1]) > 3  # the ego drives


def test_get_data_loader_batches_and_rank_shards(routes):
    class Args:
        config_path, seed, num_workers, batch_size = routes, 2021, 0, 4
    loader = datasets.get_data_loader("temporal_bev", Args)
    batch = next(iter(loader))
    assert len(loader) == 5 and [tuple(b.shape) for b in batch] == [(4, 9, 320, 320), (4, 21, 2), (4,), (4, 2), (4,), (4, 20, 21, 2), (4, 20), (4, 20), (4,)]
    assert batch[0].dtype == torch.uint8 and batch[1].dtype == torch.float64 and batch[5].dtype == torch.float32 and batch[7].dtype == torch.int32
    shards = []
    for rank in range(2):
        ld = datasets.get_data_loader("temporal_bev", Args, rank=rank, world=2)
        ld.sampler.set_epoch(3)
        assert ld.batch_size == 2
        shards.append(list(ld.sampler))
    assert not set(shards[0]) & set(shards[1]) and len(shards[0]) == len(shards[1]) == 10
    with pytest.raises(NotImplementedError, match="rgb"):
        datasets.get_data_loader("rgb", Args)


def test_train_bev_driver_reads_recorded_routes(tmp_path):
    """train_bev_v2.py without --synthetic: one epoch over a 3-frame route (one batch of 2, drop_last) on the CPU path."""
    cfg = dataset_fixture_config(str(tmp_path), routes=1, frames=23)
    out = subprocess.run([sys.executable, os.path.join(REPO, "train_bev_v2.py"), "--config-path", cfg, "--device", "cpu", "--batch-size", "2",
                          "--num-epoch", "1", "--num-workers", "0", "--save-dir", str(tmp_path / "ck")], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '"steps": 1' in out.stdout and "3 recorded frames" in out.stdout and os.path.exists(tmp_path / "ck" / "bev_1.th")
