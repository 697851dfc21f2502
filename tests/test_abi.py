"""CPU: the C-ABI library builds, loads, and exports every symbol include/ttenv.h declares."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from ddpg_trucktrailer_amd import _lib as L
    return L


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ttenv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert "tt_env_step" in names and "tt_env_create" in names and len(names) >= 15
    dll = lib.load()
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/ttenv.h but not exported by libttenv.so"
    assert sorted(lib.EXPORTS) == names, "ctypes binding and header disagree"


def test_params_default_match_reference_constants(lib):
    p = lib.default_params(0)   # simv2.py:25-57, 98-99
    assert (p.L1, p.L2, p.hitch_offset, p.v1x, p.dt) == (5.0, 7.0, 0.0, -5.012, 0.08)
    assert (p.map_min_x, p.map_max_x, p.map_min_y, p.map_max_y) == (-40.0, 40.0, -40.0, 40.0)
    assert abs(p.max_steer - 0.7853981633974483) < 1e-16 and p.position_threshold == 0.5
    assert abs(p.orientation_threshold - 0.2617993877991494) < 1e-16
    assert (p.step_length, p.extra_steps, p.fixed_max_steps, p.term_mask) == (0.40096, 75, 0, 0x3F)
    assert list(p.goal) == [0.0, -30.0, 1.5707963267948966]
    q = lib.default_params(1)   # simv1.py:34-35, 95, 432
    assert (q.L1, q.L2, q.fixed_max_steps, q.term_mask) == (5.74, 10.192, 300, 0x0F)


def test_errors_are_codes_not_exceptions(lib):
    dll = lib.load()
    h = C.c_void_p()
    assert dll.tt_env_create(0, 0, None, C.byref(h)) == lib.TT_EINVAL
    assert b"n_envs" in dll.tt_last_error(None)
    assert dll.tt_env_step(None, None, None, None, None, None, 0, None) == lib.TT_EINVAL
    assert dll.tt_env_destroy(None) == lib.TT_OK
    assert dll.tt_version() == 3


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "ddpg-trucktrailer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                for bad in ("import oracle", "from oracle", '#include "tt_oracle', "libtt_oracle", "c_oracle", "simv2_twin"):
                    assert bad not in txt, (f, bad)
