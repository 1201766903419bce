"""oracle/atari_prep.py (border-atari-env/src/env.rs:126-209 + image 0.23.14 resize, restated): properties that hold for
the published algorithm whatever the unpinned details, and hand-computed small cases of the restatement itself."""
import numpy as np

from oracle import atari_prep as AP

F = np.float32


def test_triangle_taps_hand_computed():
    # 4 -> 2: ratio 2, support 2.  o = 0: centre 1.0 -> [floor(-1) -> 0, ceil(3) = 3), centre 0.5: w = 1 - |i - 0.5| / 2
    w = AP._weights(4, 2)
    left, ws, total = w[0]
    assert left == 0 and [float(x) for x in ws] == [0.75, 0.75, 0.25] and float(total) == 1.75
    left, ws, total = w[1]
    assert left == 1 and [float(x) for x in ws] == [0.25, 0.75, 0.75] and float(total) == 1.75
    # identity size: ratio 1, support 1: taps i-1..i+1 with weights (0, 1, 0) -> the image itself
    img = np.random.default_rng(0).integers(0, 256, (84, 84, 3), dtype=np.uint8)
    assert (AP.resize_triangle(img, 84, 84) == img).all()


def test_resize_hand_computed_column():
    # one column [0, 100, 200, 40] resampled 4 -> 2 with the taps above
    col = np.array([0, 100, 200, 40], np.uint8).reshape(4, 1, 1)
    out = AP._sample_axis0(col, 2).ravel()
    a = (F(0) * F(0.75) + F(100) * F(0.75) + F(200) * F(0.25)) / F(1.75)      # 71.43 -> 71
    b = (F(100) * F(0.25) + F(200) * F(0.75) + F(40) * F(0.75)) / F(1.75)     # 117.14 -> 117
    assert out.tolist() == [int(np.floor(a + F(0.5))), int(np.floor(b + F(0.5)))] == [71, 117]


def test_constant_images_stay_constant_and_luma_quirk():
    for v in (0, 1, 127, 255):
        img = np.full((210, 160, 3), v, np.uint8)
        assert (AP.resize_triangle(img) == v).all()
    # the reference weights channel 0 with 0.114 and channel 2 with 0.299 (it calls them b and r): a pure channel-0 image
    img = np.zeros((210, 160, 3), np.uint8); img[..., 0] = 200
    assert (AP.warp_and_grayscale(img) == int(F(0.114) * F(200))).all()           # 22
    img = np.zeros((210, 160, 3), np.uint8); img[..., 2] = 200
    assert (AP.warp_and_grayscale(img) == int(F(0.299) * F(200))).all()           # 59
    white = np.full((210, 160, 3), 255, np.uint8)
    g = AP.warp_and_grayscale(white)
    assert g.min() == g.max() and g.max() in (254, 255)                            # f32 sum of the three terms, truncated


def test_resize_is_monotone_and_bounded():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 200, (210, 160, 3), dtype=np.uint8)
    b = (a + rng.integers(0, 56, a.shape, dtype=np.uint8)).astype(np.uint8)      # b >= a everywhere
    ra, rb = AP.resize_triangle(a), AP.resize_triangle(b)
    assert ra.shape == (84, 84, 3) and (rb >= ra).all()
    assert ra.min() >= a.min() and ra.max() <= a.max()


def test_frame_stack_and_clip_reward():
    rng = np.random.default_rng(2)
    fs = AP.FrameStack()
    assert (fs.frames == 0).all()
    f0 = rng.integers(0, 256, (210, 160, 3), dtype=np.uint8)
    obs = fs.reset(f0)
    g0 = AP.warp_and_grayscale(f0)
    assert all((obs[k] == g0).all() for k in range(4))
    fa, fb = rng.integers(0, 256, (2, 210, 160, 3), dtype=np.uint8)
    obs = fs.step(fa, fb)
    assert (obs[0] == AP.warp_and_grayscale(np.maximum(fa, fb))).all() and all((obs[k] == g0).all() for k in (1, 2, 3))
    fc, fd = rng.integers(0, 256, (2, 210, 160, 3), dtype=np.uint8)
    prev = obs[0].copy()
    obs = fs.step(fc, fd)
    assert (obs[1] == prev).all() and (obs[2] == g0).all() and (obs[3] == g0).all()
    assert AP.clip_reward(3.0, True) == 1.0 and AP.clip_reward(-0.5, True) == -1.0 and AP.clip_reward(0.0, True) == 0.0
    assert AP.clip_reward(3.0, False) == 3.0
