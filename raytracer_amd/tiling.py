"""Tile ownership of the multi-GPU path, host side: which tiles a rank renders and how gathered tile-major
buffers map back to the row-major frame. Mirrors rt_render_tiles_device / rt_untile_device (include/raytracer_amd.h);
the reference's model is the fixed tile grid of thread_render() (reference src/raytracer.cpp:583-617)."""
import numpy as np


def tile_grid(W, H, tw, th):
    return (W + tw - 1) // tw, (H + th - 1) // th


def tiles_owned(W, H, tw, th, part, n_parts):
    tx, ty = tile_grid(W, H, tw, th)
    n = tx * ty
    return (n - part + n_parts - 1) // n_parts if n > part else 0


def owned_tile_rects(W, H, tw, th, part, n_parts):
    """[(j, x0, y0, w, h)] for the tiles with tile_index % n_parts == part, j = ordinal in the rank's buffer."""
    tx, _ = tile_grid(W, H, tw, th)
    out = []
    for j in range(tiles_owned(W, H, tw, th, part, n_parts)):
        T = part + j * n_parts
        x0, y0 = (T % tx) * tw, (T // tx) * th
        out.append((j, x0, y0, min(tw, W - x0), min(th, H - y0)))
    return out


def untile(parts, W, H, tw, th):
    """parts[q]: uint8 array of part q's tiles (tile-major, >= tiles_owned*tw*th*3 bytes) -> (H,W,3) BGR frame."""
    n_parts = len(parts)
    frame = np.zeros((H, W, 3), np.uint8)
    for q, buf in enumerate(parts):
        tiles = np.asarray(buf, np.uint8).reshape(-1, th, tw, 3)
        for j, x0, y0, w, h in owned_tile_rects(W, H, tw, th, q, n_parts):
            frame[y0:y0 + h, x0:x0 + w] = tiles[j, :h, :w]
    return frame
