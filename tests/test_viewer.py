"""VecTask.render (reference tasks/base/vec_task.py:457-512) on the software viewer (isaacgymenvs_amd/utils/viewer.py)."""
import os

import numpy as np
import pytest


def test_render_rgb_array_shows_the_robot_and_follows_it(tmp_path):
    import isaacgymenvs_amd
    import torch
    env = isaacgymenvs_amd.make(seed=1, task="Ant", num_envs=4, sim_device="cpu", rl_device="cpu", headless=True)
    for _ in range(5):
        env.step(torch.zeros((4, 8)))
    img = env.render(mode="rgb_array")
    assert img.shape == (480, 640, 3) and img.dtype == np.uint8
    # sky at the top, checkered ground at the bottom, the robot's spheres around the image centre (the camera looks at the torso)
    sky = img[:12].reshape(-1, 3).astype(int)
    assert np.all(np.abs(sky - sky[0]).max(axis=0) <= 2) and sky[0][2] > sky[0][0]
    ground = img[-60:].reshape(-1, 3)
    assert len(np.unique(ground, axis=0)) >= 2
    centre = img[140:340, 200:440].reshape(-1, 3).astype(int)
    saturated = (centre.max(axis=1) - centre.min(axis=1)) > 60          # body colours; sky and ground are nearly grey / pale
    assert saturated.mean() > 0.01
    # the frame is a function of the state: unchanged without a step, different after the robot has moved
    again = env.render(mode="rgb_array")
    np.testing.assert_array_equal(img, again)
    for _ in range(30):
        env.step(torch.ones((4, 8)))
    moved = env.render(mode="rgb_array")
    assert (moved != img).any()
    # record_frames: frames land as PNG files named by the control step (vec_task.py:503-507)
    env.record_frames = True
    env.record_frames_dir = str(tmp_path)
    assert env.render(mode="human") is None
    f = os.path.join(str(tmp_path), f"frame_{env.control_steps}.png")
    data = open(f, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and b"IHDR" in data[:32] and data[-8:-4] == b"IEND"
    with pytest.raises(ValueError):
        env.render(mode="depth")


def test_png_writer_round_trips_through_zlib():
    import struct
    import zlib
    from isaacgymenvs_amd.utils.viewer import write_png
    import tempfile
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "a.png")
        write_png(p, img)
        data = open(p, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack(">II", data[16:24])
    assert (w, h) == (5, 7)
    i = data.index(b"IDAT")
    n = struct.unpack(">I", data[i - 4:i])[0]
    raw = zlib.decompress(data[i + 4:i + 4 + n])
    rows = np.frombuffer(raw, np.uint8).reshape(7, 1 + 15)
    assert (rows[:, 0] == 0).all()
    np.testing.assert_array_equal(rows[:, 1:].reshape(7, 5, 3), img)
