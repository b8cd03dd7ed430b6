"""Host logic of the multi-GPU frame: who owns which 64x64 tile, the single gather, and the host statement of
the tiling contract of include/rtmi.h (used by bench.py on GPUs and by the 2-rank gloo test on CPU).

New functionality (the reference is single-device): rays are independent, the scene and BVH
are replicated, rank r of N owns tiles r, r+N, ...; the only exchange step is ONE gather of shaded RGBA.
"""
from __future__ import annotations

import numpy as np

TILE = 64


def tiles_xy(resx: int, resy: int):
    return (resx + TILE - 1) // TILE, (resy + TILE - 1) // TILE


def tiling(nranks: int, root_extra: int = 0) -> int:
    """== RT_TILING: the `nranks` word of the tiled entry points (low 16 bits N, high 16 bits the extra tiles per period of
    N + e tiles that rank 0 owns; 0 = plain round robin)"""
    return int(nranks) | (int(root_extra) << 16)


def _decode(nranks: int):
    return nranks & 0xFFFF, nranks >> 16


def owned_slots(resx: int, resy: int, rank: int, nranks: int) -> int:
    """== rt_tiles_owned_pixels"""
    n, e = _decode(nranks)
    tx, ty = tiles_xy(resx, resy)
    total = tx * ty
    period = n + e
    q, rem = divmod(total, period)
    tiles = q * (e + 1) + min(rem, e + 1) if rank == 0 else q + (1 if rem > rank + e else 0)
    return tiles * TILE * TILE


def slot_map(resx: int, resy: int, nranks: int):
    """for every pixel (row-major): owning rank and slot in that rank's compact buffer"""
    n, e = _decode(nranks)
    tx, _ = tiles_xy(resx, resy)
    y, x = np.divmod(np.arange(resx * resy, dtype=np.int64), resx)
    px, py = x // TILE, y // TILE
    tile = py * tx + px
    q, p = np.divmod(tile, n + e)
    rank = np.where(p <= e, 0, p - e)
    ordinal = np.where(p <= e, q * (e + 1) + p, q)
    slot = ordinal * (TILE * TILE) + (y - py * TILE) * TILE + (x - px * TILE)
    return rank, slot


def compact_from_frame(frame: np.ndarray, resx: int, resy: int, rank: int, nranks: int, slots: int) -> np.ndarray:
    """what a rank's tiled shade kernel leaves in its compact buffer, given the full frame (host statement)"""
    r, s = slot_map(resx, resy, nranks)
    out = np.zeros((slots,) + frame.shape[1:], dtype=frame.dtype)
    mine = r == rank
    out[s[mine]] = frame[mine]
    return out


def untile_host(gathered: np.ndarray, resx: int, resy: int, nranks: int = None) -> np.ndarray:
    """== rt_untile_rgba: gathered[rank, slot] -> frame[pixel]"""
    r, s = slot_map(resx, resy, nranks if nranks is not None else gathered.shape[0])
    return gathered[r, s]


def pack_planes(rgba: np.ndarray) -> np.ndarray:
    """== what rt_shade_blinn_phong_tiled_packed writes for a compact RGBA buffer: three float planes, 12 bytes per slot
    instead of 16; alpha (1 for a hit, 0 for a miss) travels in the sign bit of the red plane (colours are in [0, 1])."""
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.ascontiguousarray(rgba[:, :3].T).copy()
    red = out[0].view(np.uint32)
    red |= np.where(rgba[:, 3] != 0.0, np.uint32(0x80000000), np.uint32(0))
    return out


def unpack_planes(planes: np.ndarray) -> np.ndarray:
    """inverse of pack_planes: [3, slots] -> [slots, 4]"""
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    bits = planes[0].view(np.uint32)
    out = np.empty((planes.shape[1], 4), dtype=np.float32)
    out[:, 0] = (bits & np.uint32(0x7FFFFFFF)).view(np.float32)
    out[:, 1] = planes[1]
    out[:, 2] = planes[2]
    out[:, 3] = np.where(bits & np.uint32(0x80000000), np.float32(1.0), np.float32(0.0))
    return out


def untile_host_packed(gathered: np.ndarray, resx: int, resy: int, nranks: int = None) -> np.ndarray:
    """== rt_untile_rgba_packed: gathered[rank, plane, slot] -> frame[pixel, rgba]"""
    r, s = slot_map(resx, resy, nranks if nranks is not None else gathered.shape[0])
    per_rank = np.stack([unpack_planes(gathered[k]) for k in range(gathered.shape[0])])
    return per_rank[r, s]


def virtual_rank(rank: int, root: int, world: int) -> int:
    """== rt_multi_frame_create_rooted: the tiles are dealt by virtual rank, the rank that assembles the frame being virtual rank 0"""
    return (rank - root + world) % world


def by_virtual_rank(gathered, root: int):
    """a buffer gathered in communicator-rank order, re-ordered by virtual rank (what the untile functions index by)"""
    world = gathered.shape[0]
    return gathered[[(v + root) % world for v in range(world)]]


def gather_to_root(send, world: int, rank: int, out=None, async_op: bool = False, root: int = 0):
    """THE collective of the path: one gather of equally sized compact buffers to the rank that assembles the frame
    (torch.distributed: RCCL over xGMI on GPUs, gloo on CPU).

    async_op=True returns (out, work): the gather runs on the communicator's own stream behind the work already queued
    on the caller's stream, so the next frame's kernels overlap it; `work.wait()` orders the caller's stream behind it."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == 'gloo' and send.is_cuda:
        # gloo has no device gather: stage through the host (rehearsals of the N > 1 path on a one-GPU box only)
        host = gather_to_root(send.cpu(), world, rank, root=root)
        if rank == root:
            if out is None:
                out = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
            out.copy_(host)

        class _Done:
            def wait(self):
                return True
        return (out if rank == root else None, _Done()) if async_op else (out if rank == root else None)
    if rank == root:
        if out is None:
            out = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        work = dist.gather(send, list(out.unbind(0)), dst=root, async_op=async_op)
    else:
        work = dist.gather(send, None, dst=root, async_op=async_op)
        out = None
    return (out, work) if async_op else out
