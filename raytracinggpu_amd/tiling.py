"""Row-tile partition of a frame over ranks and its reassembly.

Tile k (TILE_ROWS image rows) belongs to rank k mod G: contiguous blocks would be badly imbalanced (the cat
and its shadow sit in the middle rows).  Every rank renders its tiles into a dense local buffer of
`tiles_local * TILE_ROWS` rows (the last tile of some ranks may be padding); ONE gather to rank 0 per frame
moves the float4 tiles, and a single permute restores row order.  Pure torch/torch.distributed: works with
the nccl (= RCCL over xGMI) backend on GPUs and with gloo on CPU tensors.
"""
import torch
import torch.distributed as dist

TILE_ROWS = 8


def tiles_per_rank(height, world, tile_rows=TILE_ROWS):
    n_tiles = (height + tile_rows - 1) // tile_rows
    return (n_tiles + world - 1) // world


def local_buffer(height, width, world, device, tile_rows=TILE_ROWS, rgb8=False):
    """Dense per-rank buffer [tiles_local * tile_rows, W, 4] float32 (zero-filled so padding tiles are defined); rgb8: the
    tonemapped twin [.., W, 3] uint8 (float tiles for the parity check, 8-bit tiles -- 3 bytes per pixel instead of
    16 -- for the PNG path)."""
    rows = tiles_per_rank(height, world, tile_rows) * tile_rows
    if rgb8:
        return torch.zeros((rows, width, 3), dtype=torch.uint8, device=device)
    return torch.zeros((rows, width, 4), dtype=torch.float32, device=device)


def assemble(stacked, height, tile_rows=TILE_ROWS):
    """stacked: [G, tiles_local * R, W, 4] (rank-major) -> [H, W, 4].  Tile k of rank r is image tile k*G + r."""
    g, rows, w, c = stacked.shape
    t = rows // tile_rows
    return stacked.view(g, t, tile_rows, w, c).permute(1, 0, 2, 3, 4).reshape(-1, w, c)[:height]


def gather_buffer(local, world):
    """Rank 0's receive buffer [G, tiles_local * R, W, 4]: the ranks' tile buffers land in it side by side, so that
    the reassembly is ONE permuted copy (no torch.stack of separately received tensors)."""
    return torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)


def gather_frame(local, height, world, rank, gathered=None, tile_rows=TILE_ROWS, root=0):
    """One gather of the float4 tiles to rank `root` (a grouped send/recv under RCCL: each peer uses its own xGMI
    link to the root).  `gathered`: the root's gather_buffer (allocated here if None).  Returns the assembled
    [H, W, 4] frame on the root, None elsewhere.  A caller that renders frame after frame may ROTATE the root
    (frame k -> rank k mod G): every frame is still one gather into one device, but the inbound traffic and the
    assembly are spread over all ranks' links instead of loading rank 0's seven."""
    if world == 1:
        return local[:height]
    if rank == root and gathered is None:
        gathered = gather_buffer(local, world)
    dist.gather(local, list(gathered.unbind(0)) if rank == root else None, dst=root)
    if rank != root:
        return None
    return assemble(gathered, height, tile_rows)


def root_of(frame_index, world, policy="0"):
    """Which rank assembles frame `frame_index`: "0" = always rank 0 (the reference copies every image to ONE host, optimized.cu:849-856); "rotate" = rank k mod G."""
    return frame_index % world if policy == "rotate" else 0
