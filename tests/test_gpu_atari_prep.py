"""(f)-4: device-side Atari frame preprocessing vs oracle/atari_prep.py - bit-exact (byte work).  The oracle restates
border-atari-env/src/env.rs:126-209 + image 0.23.14's Triangle resize; it is unpinned against the real crate (no vector
obtainable offline), so what is proved here is kernel == restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() < 1:
        pytest.skip("no GPU")
    return border_amd


def frames(rng, n, kind):
    if kind == "noise":
        return rng.integers(0, 256, (n, 210, 160, 3), dtype=np.uint8)
    # game-like: flat background, a few bright rectangles
    f = np.full((n, 210, 160, 3), (20, 40, 90), np.uint8)
    for k in range(n):
        for _ in range(6):
            y, x = rng.integers(0, 200), rng.integers(0, 150)
            f[k, y:y + rng.integers(2, 30), x:x + rng.integers(2, 30)] = rng.integers(0, 256, 3)
    return f


@pytest.mark.parametrize("kind", ["noise", "game"])
def test_reset_and_steps_match_the_oracle_bit_for_bit(B, kind):
    from oracle import atari_prep as AP
    rng = np.random.default_rng(7)
    n_envs = 5
    prep = B.AtariPreprocessor(n_envs)
    assert (prep.obs(range(n_envs)) == 0).all()                       # frames: vec![0; 4*84*84]
    stacks = [AP.FrameStack() for _ in range(n_envs)]
    f0 = frames(rng, n_envs, kind)
    got = prep.reset(range(n_envs), f0)
    for e in range(n_envs):
        assert (got[e] == stacks[e].reset(f0[e])).all(), e
    for step in range(6):
        envs = [e for e in range(n_envs) if (step + e) % 3 != 0] or [0]        # ragged subsets, any order
        envs = envs[::-1] if step % 2 else envs
        fa, fb = frames(rng, len(envs), kind), frames(rng, len(envs), kind)
        got = prep.step(envs, fa, fb)
        for k, e in enumerate(envs):
            want = stacks[e].step(fa[k], fb[k])
            assert (got[k] == want).all(), (step, e, np.abs(got[k].astype(int) - want.astype(int)).max())
    # environments that did not step kept their stacks
    all_obs = prep.obs(range(n_envs))
    for e in range(n_envs):
        assert (all_obs[e] == stacks[e].frames).all()
    # a mid-episode reset of one environment
    f1 = frames(rng, 1, kind)
    assert (prep.reset([3], f1)[0] == stacks[3].reset(f1[0])).all()
    prep.close()


def test_other_frame_sizes_and_argument_checks(B):
    from oracle import atari_prep as AP
    rng = np.random.default_rng(9)
    prep = B.AtariPreprocessor(2, width=160, height=250)     # PAL-sized ALE screens
    f = rng.integers(0, 256, (2, 250, 160, 3), dtype=np.uint8)
    got = prep.reset([0, 1], f)
    for e in range(2):
        assert (got[e][0] == AP.warp_and_grayscale(f[e])).all()
    with pytest.raises(B.BdrError):
        prep.step([0, 0], f, f)              # an environment twice in one call
    with pytest.raises(B.BdrError):
        prep.step([0, 2], f, f)              # out of range
    prep.close()
    small = B.AtariPreprocessor(1, width=84, height=84)      # identity resize
    img = rng.integers(0, 256, (1, 84, 84, 3), dtype=np.uint8)
    assert (small.reset([0], img)[0][0] == AP.grayscale(img[0])).all()
    small.close()
    assert B.AtariPreprocessor.clip_reward(7.0, True) == 1.0 and B.AtariPreprocessor.clip_reward(-7.0, False) == -7.0
