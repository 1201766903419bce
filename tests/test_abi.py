"""The C-ABI library builds for gfx950, loads, and exports every symbol include/border_amd.h
declares.  No compute calls: this runs in the CPU-only container."""
import ctypes as C
import os
import re

import pytest

from border_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    return build.build_library()


def test_header_symbols_are_exported(libpath):
    hdr = open(os.path.join(ROOT, "include", "border_amd.h")).read()
    declared = sorted(set(re.findall(r"BDR_API[^;(]*?\b(bdr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    L = C.CDLL(libpath)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(declared) == sorted(_lib.ABI_SYMBOLS)


def test_loader_and_struct_layouts(libpath):
    L = _lib.lib()
    assert b"gfx950" in L.bdr_version()
    c = _lib.DqnConfigC()
    L.bdr_dqn_config_default(C.byref(c))
    # dqn/config.rs:82-102
    assert (c.soft_update_interval, c.n_updates_per_opt, c.batch_size) == (1, 1, 1)
    assert c.discount_factor == 0.99 and c.tau == 0.005 and c.critic_loss == 0 and c.double_dqn == 0
    assert c.device == -1 and c.net.n_stack == 4


def test_no_device_fails_loudly(libpath):
    """No silent CPU fallback: without a GPU, constructors return an error."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    cfg = _lib.ReplayConfig(16, 42, 16, 8, 0, 0, 0)
    h = C.c_void_p()
    assert _lib.lib().bdr_replay_create(C.byref(cfg), C.byref(h)) == 2  # BDR_ERR_NO_DEVICE
    with pytest.raises(_lib.BdrError):
        from border_amd import SimpleReplayBuffer, SimpleReplayBufferConfig
        SimpleReplayBuffer(SimpleReplayBufferConfig(capacity=16), (4,), "float32")
