"""The C ABI library loads and exports every symbol the header declares (no GPU needed)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mppi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from trajopt_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names
    assert L.mppi_version() == 8


def test_struct_layouts_match_the_header():
    from trajopt_amd import _lib
    # sizes implied by the header's field lists (int32 / double, natural alignment)
    nd = 32
    model = 64 + 4 * 2 + 4 * nd * 2 + 4 * 2 + 8 + 8 * (3 * nd * 3 + 6 * nd + 9 * nd) + 8 * (3 + 4 + 3 + 2 + 5) + 8 * 3 + 4 * nd + 4 + 4 * nd + 4
    assert ctypes.sizeof(_lib.ModelDescC) == model
    assert ctypes.sizeof(_lib.PlannerCfgC) == 4 * 10 + 8 * 2 + 8 * (nd + 3 + nd)


def test_philox_known_answers():
    """Philox-4x32-10 vectors of the Random123 distribution's known-answer file."""
    from trajopt_amd import _lib
    L = _lib.lib()
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        c, k, o = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
        L.mppi_philox4x32(c.ctypes.data, k.ctypes.data, o.ctypes.data)
        assert [int(x) for x in o] == want


def test_model_binding_rejects_a_model_without_kernel():
    import copy
    import pytest
    from trajopt_amd import _lib
    from trajopt_amd.model import get_model
    md = copy.deepcopy(get_model("reacher_7dof"))
    md.damping[3] = 0.5
    h = _lib._vp()
    c = _lib.desc_to_c(md)
    rc = _lib.lib().mppi_model_create(ctypes.byref(c), ctypes.byref(h))
    assert rc != 0 and b"damping" in _lib.lib().mppi_last_error()
    with pytest.raises(_lib.MppiError):
        _lib.check(rc)
    md.name = "no_such_model"
    c = _lib.desc_to_c(md)
    assert _lib.lib().mppi_model_create(ctypes.byref(c), ctypes.byref(h)) != 0
