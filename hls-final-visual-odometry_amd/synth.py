"""Deterministic KITTI-shaped synthetic frames.

The reference ships no input data (its .dat frames are git-ignored, reference
.gitignore:6), so every test and the benchmark use this generator:

  canvas (W+64)x(H+64) of LCG noise  ->  `blur` passes of a 3x3 integer box
  blur  ->  gain  ->  crop at offset (32+dx, 32+dy).

Frames of one sequence share the canvas and differ only in (dx,dy) (an integer
pan, so the true flow is known); the right camera is the left one panned by a
constant disparity.  Padding columns [W,bpl) are zero.
"""
from __future__ import annotations

import functools

import numpy as np

_LCG_A = 1664525
_LCG_C = 1013904223


def bytes_per_line(width: int) -> int:
    """dims[2] as Matcher::pushBack pads it (reference src/matcher.cpp:84)."""
    return width + 15 - (width - 1) % 16


def _lcg_bytes(n: int, seed: int) -> np.ndarray:
    """base[i] = s_{i+1} >> 24 with s_{k+1} = s_k*A + C (mod 2^32), vectorised."""
    a = np.full(n, _LCG_A, dtype=np.uint32)
    apow = np.cumprod(a, dtype=np.uint32)              # A^(i+1)
    geo = np.empty(n, dtype=np.uint32)                 # 1 + A + ... + A^i
    geo[0] = 1
    if n > 1:
        geo[1:] = np.cumsum(apow[:-1], dtype=np.uint32) + np.uint32(1)
    s = apow * np.uint32(seed & 0xFFFFFFFF) + np.uint32(_LCG_C) * geo
    return (s >> np.uint32(24)).astype(np.uint8)


@functools.lru_cache(maxsize=8)
def canvas(width: int, height: int, blur: int = 8, seed: int = 1) -> np.ndarray:
    """Blurred noise canvas of shape (H+64, W+64), uint8."""
    wc, hc = width + 64, height + 64
    base = _lcg_bytes(wc * hc, seed).reshape(hc, wc)
    for _ in range(blur):
        t = base.astype(np.int32)
        acc = np.zeros((hc - 2, wc - 2), dtype=np.int32)
        for dy in range(3):
            for dx in range(3):
                acc += t[dy:dy + hc - 2, dx:dx + wc - 2]
        base = base.copy()
        base[1:-1, 1:-1] = (acc // 9).astype(np.uint8)
    base.setflags(write=False)
    return base


def frame(width: int, height: int, dx: int = 0, dy: int = 0, blur: int = 8,
          gain: int = 1, seed: int = 1, bpl: int | None = None) -> np.ndarray:
    """One frame as a (H, bpl) uint8 array; columns >= W are zero."""
    if bpl is None:
        bpl = bytes_per_line(width)
    if not (-32 <= dx <= 32 and -32 <= dy <= 32):
        raise ValueError("pan must stay inside the 32-pixel canvas border")
    cv = canvas(width, height, blur, seed)
    crop = cv[32 + dy:32 + dy + height, 32 + dx:32 + dx + width].astype(np.int32)
    img = np.zeros((height, bpl), dtype=np.uint8)
    img[:, :width] = np.clip((crop - 128) * gain + 128, 0, 255).astype(np.uint8)
    return img


def stereo_sequence(width: int, height: int, n_frames: int, disparity: int = 12,
                    blur: int = 8, gain: int = 1, seed: int = 1):
    """[(left_t, right_t)] with left pan (5t mod 32, t mod 32); right = left
    panned by a further +disparity pixels (content shifts left in the right
    image, i.e. u_left >= u_right as a rectified stereo rig gives)."""
    out = []
    for t in range(n_frames):
        dx, dy = (5 * t) % 20, t % 20
        left = frame(width, height, dx, dy, blur, gain, seed)
        right = frame(width, height, dx + disparity, dy, blur, gain, seed)
        out.append((left, right))
    return out


def fnv1a64(buf) -> int:
    """FNV-1a-64 over raw bytes (slow pure-numpy fold; for small checks use the
    oracle's C version)."""
    h = 1469598103934665603
    for b in np.frombuffer(np.ascontiguousarray(buf).tobytes(), dtype=np.uint8):
        h = ((h ^ int(b)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h
