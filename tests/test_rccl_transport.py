"""The RCCL tile gather of the one-process-per-GPU path (include/raytrace_rccl.h, libraytrace_rccl.so).

CPU: the library loads, exports what the header declares, refuses bad arguments, and its tile plan -- the offsets and sizes
rt_comm_gather_tiles sends and receives with -- replayed over numpy buffers rebuilds the frame for every world size.
GPU (-m gpu): a one-rank communicator through the real library (RCCL refuses two ranks on one device, which is tested too), the
render -> tonemap -> gather chain on the communicator's stream, and the C++ launcher's --rccl-id path.  More than one rank of
RCCL needs more than one GPU: that leg has not run on hardware."""
import os
import re
import subprocess
import time
import sys

import numpy as np
import pytest

import raytracinggpu_amd as rt
from raytracinggpu_amd import _rccl
from tests.conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "raytrace_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(rt_comm_[a-z0-9_]+)\s*\(", text)))
    lib = _rccl.load()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_rccl.EXPORTS) == names
    assert lib.rt_comm_abi_version() == 2


@pytest.mark.parametrize("W,H,bpp,tile_rows,world", [(400, 250, 16, 8, 1), (400, 250, 16, 8, 2), (400, 250, 3, 8, 3), (1920, 1080, 3, 8, 8),
                                                     (64, 7, 16, 8, 4), (33, 130, 16, 16, 5), (7680, 4320, 3, 8, 8)])
def test_tile_plan_replayed_rebuilds_the_frame(W, H, bpp, tile_rows, world):
    """What rank r sends and the root receives, replayed on the host: every rank's dense tile buffer (the rows interleaved_rows /
    rt_rows name, raytrace_hip.h) lands in place, every byte of the frame is written exactly once, and the k-th send of a rank
    matches the root's k-th receive from it in size."""
    rng = np.random.default_rng(W * 31 + H)
    row_bytes = W * bpp
    frame = rng.integers(0, 256, size=(H, row_bytes), dtype=np.uint8)
    local = []
    for rank in range(world):
        rows, idx = rt.interleaved_rows(H, tile_rows, rank, world)
        assert rows.n_rows == len(idx)
        local.append(frame[idx].reshape(-1))
    out = np.zeros(H * row_bytes, np.uint8)
    written = np.zeros(H * row_bytes, np.uint8)
    n_tiles = (H + tile_rows - 1) // tile_rows
    sends = {r: [] for r in range(world)}
    recvs = {r: [] for r in range(world)}
    for t in range(n_tiles):
        p = _rccl.tile_plan(W, H, bpp, tile_rows, world, t)
        assert p.owner == t % world and p.bytes == p.rows * row_bytes and p.frame_offset == t * tile_rows * row_bytes
        assert p.local_offset + p.bytes <= local[p.owner].size
        out[p.frame_offset:p.frame_offset + p.bytes] = local[p.owner][p.local_offset:p.local_offset + p.bytes]
        written[p.frame_offset:p.frame_offset + p.bytes] += 1
        sends[p.owner].append(p.bytes)       # issue order on the owner: increasing t
        recvs[p.owner].append(p.bytes)       # issue order on the root: increasing t
    assert (written == 1).all()
    np.testing.assert_array_equal(out.reshape(H, row_bytes), frame)
    assert sends == recvs
    assert sum(len(v) for v in sends.values()) == n_tiles
    with pytest.raises(_rccl.CommError):
        _rccl.tile_plan(W, H, bpp, tile_rows, world, n_tiles)


@pytest.mark.parametrize("W,H,bpp,tile_rows", [(400, 250, 16, 8), (400, 250, 3, 8), (1920, 1080, 3, 8), (64, 7, 16, 8), (33, 130, 3, 16), (101, 57, 1, 8)])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_coalesced_plan_replayed_rebuilds_the_frame(W, H, bpp, tile_rows, world):
    """The second exchange plan, replayed on the host for world 1...8 and both roots that matter: every peer's
    ONE message -- its whole dense buffer -- lands at its 256-byte aligned place in the root's staging area, the placement pass (what
    place_tiles_kernel does, word by word) copies tile t from stage(owner) + tile_plan(t).local_offset, the root's own tiles from its
    own buffer; the frame comes back whole, pieces of the staging area never overlap, ranks without tiles send nothing."""
    rng = np.random.default_rng(W * 7 + H * 3 + world)
    row_bytes = W * bpp
    frame = rng.integers(0, 256, size=(H, row_bytes), dtype=np.uint8)
    n_tiles = (H + tile_rows - 1) // tile_rows
    for root in sorted({0, world - 1}):
        local = []
        for rank in range(world):
            _, idx = rt.interleaved_rows(H, tile_rows, rank, world)
            local.append(frame[idx].reshape(-1))
        plans = [_rccl.peer_plan(W, H, bpp, tile_rows, world, root, r) for r in range(world)]
        end = 0
        for r in range(world):
            pp = plans[r]
            assert pp.bytes == local[r].size and pp.n_tiles == len(range(r, n_tiles, world))
            if r == root:
                assert pp.stage_offset == 0
                continue
            assert pp.stage_offset % 256 == 0 and pp.stage_offset >= end       # pieces in rank order, none overlapping
            end = pp.stage_offset + pp.bytes
        stage = np.full(end + 256, 0xEE, np.uint8)
        messages = 0
        for r in range(world):                                                  # the exchange: one message per peer that holds tiles
            if r != root and plans[r].bytes:
                stage[plans[r].stage_offset:plans[r].stage_offset + plans[r].bytes] = local[r]
                messages += 1
        assert messages == sum(1 for r in range(world) if r != root and r < n_tiles)
        out = np.zeros(H * row_bytes, np.uint8)
        for t in range(n_tiles):                                                # the placement pass
            p = _rccl.tile_plan(W, H, bpp, tile_rows, world, t)
            src = local[root] if p.owner == root else stage[plans[p.owner].stage_offset:]
            out[p.frame_offset:p.frame_offset + p.bytes] = src[p.local_offset:p.local_offset + p.bytes]
        np.testing.assert_array_equal(out.reshape(H, row_bytes), frame)
    with pytest.raises(_rccl.CommError):
        _rccl.peer_plan(W, H, bpp, tile_rows, world, 0, world)


def test_auto_plan_follows_the_tile_size():
    """AUTO = coalesced below the byte threshold, per tile at or above it; an explicit plan is never overridden."""
    assert _rccl.choose_plan("auto", 1920, 3, 8) == "coalesced"                # 46 KB tiles (1080p RGB8)
    assert _rccl.choose_plan("auto", 1920, 16, 8) == "coalesced"               # 245 KB (1080p float4): still below 256 KiB
    assert _rccl.choose_plan("auto", 7680, 16, 8) == "tile"                    # 983 KB (config 5, float4)
    assert _rccl.choose_plan("auto", 7680, 3, 8) == "coalesced"                # 184 KB (config 5, RGB8)
    assert _rccl.choose_plan("auto", 7680, 3, 8, coalesce_below=100 * 1024) == "tile"
    assert _rccl.choose_plan("tile", 64, 3, 8) == "tile" and _rccl.choose_plan("coalesced", 7680, 16, 8) == "coalesced"


def test_bad_arguments_and_no_gpu_fail_loudly():
    L = _rccl.load()
    import ctypes as C
    h = C.c_void_p()
    assert L.rt_comm_create(C.byref(h), 0, 0, 0, b"\0" * 128) == -1        # world 0
    assert L.rt_comm_create(C.byref(h), 0, 2, 2, b"\0" * 128) == -1        # rank outside the world
    assert L.rt_comm_gather_tiles(None, None, 4, 4, 16, 8, 0, None, None) == -1
    assert b"NULL" in L.rt_comm_last_error(None)
    if rt.device_count() == 0:
        with pytest.raises(_rccl.CommError):
            _rccl.Comm(0, 0, 1, b"\0" * 128)                                # no device: an error, not another transport


# ------------------------------------------------------------------ GPU ------------------------------------------------------------------

def _upload_cat(ctx, cat_golden):
    ctx.scene_upload(rt.scenes.spheres("cpu"), dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"],
                                                    albedo=rt.scenes.CAT_ALBEDO, object_slot=6))


@pytest.mark.gpu
def test_one_rank_gather_on_the_communicator_stream_is_the_frame(cat_golden):
    """render -> (tonemap) -> rt_comm_gather_tiles, all on the communicator's stream, no host wait in between: the frame of rt_render
    / rt_render_rgb8 bit for bit (250 rows: the last tile is a short one)."""
    import torch
    ctx = rt.Context(0)
    _upload_cat(ctx, cat_golden)
    W, H = 400, 250
    p = rt.make_params(W, H, 2, 2, **rt.scenes.CPU_LAUNCHER)
    full = ctx.render(p)
    full8 = ctx.render_rgb8(p)
    comm = _rccl.Comm(0, 0, 1, _rccl.unique_id())
    rows, idx = rt.interleaved_rows(H, 8, 0, 1)
    tiles = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    tiles8 = torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device="cuda:0")
    frame8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ctx.render_device(p, rows, tiles.data_ptr(), stream=comm.stream)
    comm.gather_tiles(tiles.data_ptr(), W, H, 16, frame.data_ptr())
    ctx.tonemap_device(tiles.data_ptr(), H * W, tiles8.data_ptr(), stream=comm.stream)
    comm.gather_tiles(tiles8.data_ptr(), W, H, 3, frame8.data_ptr())
    comm.sync()
    np.testing.assert_array_equal(frame.cpu().numpy().view(np.uint32), full.view(np.uint32))
    np.testing.assert_array_equal(frame8.cpu().numpy(), full8)
    assert comm.last_bytes == 0                                            # nothing crossed the fabric: the root's own tiles are device copies
    with pytest.raises(_rccl.CommError):
        comm.gather_tiles(tiles.data_ptr(), W, H, 16, None)                # the root must name its frame
    comm.close()
    ctx.close()


_TWO_RANKS = r"""
import os, sys, time
sys.path.insert(0, %(root)r)
from raytracinggpu_amd import _rccl
rank, idfile = int(sys.argv[1]), sys.argv[2]
if rank == 0:
    open(idfile + ".tmp", "wb").write(_rccl.unique_id()); os.rename(idfile + ".tmp", idfile)
for _ in range(600):
    if os.path.exists(idfile): break
    time.sleep(0.1)
try:
    _rccl.Comm(0, rank, 2, open(idfile, "rb").read())
    print("CREATED")
except _rccl.CommError as e:
    print("REFUSED", e.code)
"""


@pytest.mark.gpu
def test_two_ranks_on_one_device_are_refused_not_hung(tmp_path):
    """RCCL does not let two ranks of a communicator share a device: rt_comm_create reports RT_COMM_ERR_RCCL on both (and returns)."""
    if rt.device_count() > 1:
        pytest.skip("more than one GPU: RCCL accepts the ranks on distinct devices -- tests/test_multi_rank.py::test_two_real_rccl_ranks_* run them")
    script = tmp_path / "two.py"
    script.write_text(_TWO_RANKS % {"root": ROOT})
    ps = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / "id")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in ps:
            outs.append(p.communicate(timeout=180)[0])
    finally:
        for p in ps:                                                    # a rank that did hang must not outlive the test
            if p.poll() is None:
                p.kill()
                p.communicate()
    for out in outs:
        assert "REFUSED -3" in out, out


@pytest.mark.gpu
def test_launcher_rccl_path_writes_the_reference_png(tmp_path, cat_golden):
    """`rt_launcher 1 0 --tile-rank 0 --tile-world 1 --rccl-id FILE`: the C++ process-per-GPU path with its RCCL exchange
    (Renderer::render_gather_rgb8 + TileComm, the library loaded on demand) == the bytes of the reference's `./cpu 1 0`."""
    from PIL import Image
    launcher = os.path.join(ROOT, "raytracinggpu_amd", "rt_launcher")
    g = load_golden("ref_cpu_png_1_0.npz")
    d = tmp_path / "cadnav.com_model" / "Models_F0202A090"
    d.mkdir(parents=True)
    with open(d / "cat.obj", "w") as f:
        for v in cat_golden["vertices"]:
            f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
        for t in cat_golden["tri_obj_order"]:
            f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
    (tmp_path / "id").write_bytes(b"x" * 128)                              # what an aborted earlier run may have left at the path: rank 0 removes it first
    r = subprocess.run([launcher, "1", "0", "--tile-rank", "0", "--tile-world", "1", "--rccl-id", str(tmp_path / "id"), "--out", "rccl.png"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Rendering time: " in r.stdout                                  # (RCCL prints its version banner to stdout first)
    assert "over RCCL" in r.stderr
    assert not (tmp_path / "id").exists()                                  # ... and leaves none behind once the communicator is up
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "rccl.png").convert("RGB")), g["cat"])


@pytest.mark.gpu
def test_launcher_rejects_a_stale_communicator_id(tmp_path):
    """a rank other than 0 used to take ANY 128-byte file at the --rccl-id path for this run's id and then block for ever
    in the collective set-up next to ranks holding another id.  The file now carries the launch's nonce: a leftover of another launch
    (raw 128 bytes as the old format wrote them, or a well-formed file of another nonce) is ignored, the wait for the real one is
    bounded by --rccl-timeout, and the process ends with an error instead of hanging with the GPU initialised."""
    launcher = os.path.join(ROOT, "raytracinggpu_amd", "rt_launcher")
    for stale in (b"\x01" * 128, b"rtid:someone-else:" + b"\x02" * 128):
        (tmp_path / "id").write_bytes(stale)
        t0 = time.time()
        r = subprocess.run([launcher, "1", "0", "--scene", "spheres", "--tile-rank", "1", "--tile-world", "2", "--rccl-id", str(tmp_path / "id"),
                            "--rccl-nonce", "this-launch", "--rccl-timeout", "2", "--width", "64", "--height", "64"],
                           cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert r.returncode == 1, (r.returncode, r.stderr)
        assert "no communicator id of launch this-launch" in r.stderr
        assert time.time() - t0 < 60
        assert (tmp_path / "id").read_bytes() == stale                     # only rank 0 ever removes the file
