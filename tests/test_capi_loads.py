"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, and refuses to
work without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

import raytracinggpu_amd as rt
from raytracinggpu_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raytrace_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_capi.EXPORTS) == names
    assert lib.rt_abi_version() == 6


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_capi.Sphere) == 40
    assert ctypes.sizeof(_capi.Mesh) == 72                              # ABI 6: + mirror, in / out refraction index (Geometry, cpu:113-116)
    assert ctypes.sizeof(_capi.Params) == 40
    assert ctypes.sizeof(_capi.Rows) == 16
    assert ctypes.sizeof(_capi.FrameDesc) == 32
    assert ctypes.sizeof(_capi.Light) == 16 and ctypes.sizeof(_capi.Camera) == 16
    assert ctypes.sizeof(_capi.Stats) == 64 and ctypes.sizeof(_capi.Work) == 32 + 16 + 12 * 8 and ctypes.sizeof(_capi.CameraPose) == 24 and ctypes.sizeof(_capi.KatCounts) == 40
    assert ctypes.sizeof(_capi.MultiStats) == 4 + 64 + 64 + 4 + 4 + 4 + 8 + 8 + 64 + 4 + 4     # incl. 4 bytes of padding before `rays`, submit_ms + tail padding


def test_no_cpu_fallback_without_a_gpu():
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Context(0)
    assert e.value.code in (-2, -3)


def test_camera_basis_matches_oracle_restatement(oracle):
    """rt_camera_basis is host code (Camera::rotate(), realtime_render.cu:823-846): identical to the oracle's."""
    for yaw, pitch in ((0.0, 0.3), (0.7, -0.2), (-2.5, 1.1), (3.0, 0.0)):
        got = rt.camera_basis(rt.make_pose(yaw=yaw, pitch=pitch))
        exp = oracle.camera_basis(yaw, pitch)
        for a, b in zip(got, exp):
            np.testing.assert_array_equal(a, b)
        m = np.stack(got).astype(np.float64)
        np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-6)   # orthonormal


def test_multi_device_entry_fails_loudly_without_a_gpu():
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError):
        rt.MultiContext([0, 0])
    with pytest.raises(rt.RtError):
        rt.MultiContext([])


def test_interleaved_rows_partition_the_image():
    for H, R, G in ((1080, 8, 8), (1080, 8, 3), (250, 8, 4), (7, 8, 2), (4320, 16, 8)):
        seen = np.zeros(H, int)
        for k in range(G):
            rows, idx = rt.interleaved_rows(H, R, k, G)
            assert rows.n_rows == len(idx)
            local = np.arange(rows.n_rows)
            mapped = rows.row0 + (local // rows.tile_rows) * rows.tile_rows * rows.tile_step + local % rows.tile_rows
            np.testing.assert_array_equal(mapped, idx)      # the formula documented in raytrace_hip.h
            seen[idx] += 1
        assert (seen == 1).all()
