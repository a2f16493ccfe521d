"""Deterministic synthetic 8-bit 4:2:0 sequences.

No files, no network: a 32-bit LCG (s = s*1664525 + 1013904223, seed 12345)
drives everything, evaluated with numpy's wrapping uint32 arithmetic so that an
8K frame takes a fraction of a second.

Frame 0 luma:  Y0[y][x] = u8(((3x + 2y) >> 2) + ((s >> 24) & 15))   (gradient + 4-bit noise)
Frame n > 0 :  frame n-1 translated (wrap-around) by a per-frame vector drawn
               from the LCG in [-R/2, R/2]^2, plus fresh 2-bit noise, plus two
               128x128 textured rectangles moving with their own vectors -- so
               the true motion is non-trivial, inside the search range, and
               differs between CTUs.
Chroma      :  128 + ((s >> 24) & 7), translated by half the luma vector.
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


class LCG:
    """Vectorised LCG stream: take(n) returns the next n states (after stepping)."""

    def __init__(self, seed=12345):
        self.s = np.uint32(seed)

    def take(self, n):
        n = int(n)
        if n == 0:
            return np.zeros(0, np.uint32)
        with np.errstate(over="ignore"):
            # a^k and c * (1 + a + ... + a^(k-1)) for k = 1..n, all mod 2^32
            ak = np.full(n, _A, np.uint32)
            np.cumprod(ak, out=ak, dtype=np.uint32)
            geo = np.empty(n, np.uint32)
            geo[0] = 1
            if n > 1:
                geo[1:] = ak[:-1]
            np.cumsum(geo, out=geo, dtype=np.uint32)
            out = ak * self.s + geo * _C
        self.s = out[-1]
        return out

    def next_int(self, lo, hi):
        """One draw, uniform-ish integer in [lo, hi]."""
        v = int(self.take(1)[0] >> np.uint32(16))
        return lo + v % (hi - lo + 1)


def pad_plane(p, mult):
    """Edge-replicate a plane to multiples of `mult` (1080 -> 1088; src/params.cpp:73-74
    rejects sizes that are not a multiple of the CU size)."""
    h, w = p.shape
    H = (h + mult - 1) // mult * mult
    W = (w + mult - 1) // mult * mult
    if (H, W) == (h, w):
        return p
    return np.pad(p, ((0, H - h), (0, W - w)), mode="edge")


def make_sequence(width, height, nframes, search_range=64, seed=12345, chroma=False):
    """Returns luma uint8[nframes][height][width] (and (U, V) if chroma=True).

    width/height are the size actually generated (pass the padded size to get
    content out to the padded edge, or pad a nominal-size sequence with
    pad_plane())."""
    g = LCG(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.uint32)
    base = ((np.uint32(3) * xx + np.uint32(2) * yy) >> np.uint32(2))
    noise = (g.take(width * height) >> np.uint32(24)) & np.uint32(15)
    y0 = ((base + noise.reshape(height, width)) & np.uint32(255)).astype(np.uint8)
    luma = np.empty((nframes, height, width), np.uint8)
    luma[0] = y0
    cw, ch = width // 2, height // 2
    if chroma:
        u = np.empty((nframes, ch, cw), np.uint8)
        v = np.empty((nframes, ch, cw), np.uint8)
        u[0] = (128 + ((g.take(cw * ch) >> np.uint32(24)) & np.uint32(7))).astype(np.uint8).reshape(ch, cw)
        v[0] = (128 + ((g.take(cw * ch) >> np.uint32(24)) & np.uint32(7))).astype(np.uint8).reshape(ch, cw)
    # two textured rectangles with their own motion
    rs = min(128, width // 2, height // 2)
    rects = []
    for _ in range(2):
        tex = (g.take(rs * rs) >> np.uint32(24)).astype(np.uint8).reshape(rs, rs)
        px = g.next_int(0, width - rs)
        py = g.next_int(0, height - rs)
        vx = g.next_int(-search_range // 4, search_range // 4)
        vy = g.next_int(-search_range // 4, search_range // 4)
        rects.append([tex, px, py, vx, vy])
    half = search_range // 2
    for n in range(1, nframes):
        gx = g.next_int(-half, half)
        gy = g.next_int(-half, half)
        f = np.roll(luma[n - 1], (gy, gx), axis=(0, 1)).astype(np.uint32)
        f += (g.take(width * height) >> np.uint32(24)).reshape(height, width) & np.uint32(3)
        f = (f & np.uint32(255)).astype(np.uint8)
        for r in rects:
            tex, px, py, vx, vy = r
            x = (px + n * vx) % (width - rs + 1)
            y = (py + n * vy) % (height - rs + 1)
            f[y:y + rs, x:x + rs] = tex
        luma[n] = f
        if chroma:
            u[n] = np.roll(u[n - 1], (gy // 2, gx // 2), axis=(0, 1))
            v[n] = np.roll(v[n - 1], (gy // 2, gx // 2), axis=(0, 1))
    if chroma:
        return luma, u, v
    return luma


def make_textured_sequence(width, height, nframes, search_range=64, ctu=64, seed=777):
    """A third content class beside the smooth sequence and white noise: 1/f-like texture (seven octaves of LCG
    value noise from 128- to 2-pixel cells, bilinearly interpolated, amplitude x 0.7 per octave) in which EVERY ctu x ctu block of a frame
    moves with its own vector -- frame n is cut from frame 0 block by block, the vectors drawn from the LCG in
    [-R/2, R/2]^2 (clamped to the picture), plus fresh 2-bit noise.  Bounds prune part of the candidates here,
    neither nearly all (the smooth sequence) nor none (white noise).  Returns luma uint8[nframes][height][width]."""
    g = LCG(seed)
    acc = np.zeros((height, width), np.float32)
    amp = 48.0
    for octave in range(7):
        cell = 128 >> octave                        # 128, 64, .. 2 pixels per lattice cell
        gh, gw = height // cell + 2, width // cell + 2
        lat = ((g.take(gh * gw) >> np.uint32(24)).astype(np.float32) / 255.0 - 0.5).reshape(gh, gw)
        yy = np.arange(height, dtype=np.float32) / cell
        xx = np.arange(width, dtype=np.float32) / cell
        y0, x0 = yy.astype(np.int32), xx.astype(np.int32)
        fy, fx = (yy - y0)[:, None], (xx - x0)[None, :]
        a = lat[y0][:, x0] * (1 - fx) + lat[y0][:, x0 + 1] * fx
        b = lat[y0 + 1][:, x0] * (1 - fx) + lat[y0 + 1][:, x0 + 1] * fx
        acc += amp * (a * (1 - fy) + b * fy)
        amp *= 0.7
    base = np.clip(np.rint(acc + 128.0), 0, 255).astype(np.uint8)
    luma = np.empty((nframes, height, width), np.uint8)
    luma[0] = base
    half = search_range // 2
    for n in range(1, nframes):
        f = np.empty((height, width), np.uint32)
        for by in range(0, height, ctu):
            for bx in range(0, width, ctu):
                bh, bw = min(ctu, height - by), min(ctu, width - bx)
                dx = min(max(bx + g.next_int(-half, half), 0), width - bw) - bx
                dy = min(max(by + g.next_int(-half, half), 0), height - bh) - by
                f[by:by + bh, bx:bx + bw] = base[by + dy:by + dy + bh, bx + dx:bx + dx + bw]
        f += (g.take(width * height) >> np.uint32(24)).reshape(height, width) & np.uint32(3)
        luma[n] = np.minimum(f, 255).astype(np.uint8)
    return luma


def make_intra_probe(width, height, nframes, seed=12345):
    """The SURVEY.md App. B whole-encoder probe input (BASELINE.json configs[0]):
    LCG state carried across frames; Y = u8(((x+2n)*3 + (y+n)*2)/4 + ((s>>24)&15)),
    chroma 128 + ((s>>24)&7).  Returns uint8[nframes][width*height*3/2]."""
    g = LCG(seed)
    out = np.empty((nframes, width * height * 3 // 2), np.uint8)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.uint32)
    for n in range(nframes):
        base = ((xx + np.uint32(2 * n)) * np.uint32(3) + (yy + np.uint32(n)) * np.uint32(2)) // np.uint32(4)
        ny = (g.take(width * height) >> np.uint32(24)) & np.uint32(15)
        out[n, :width * height] = ((base.ravel() + ny) & np.uint32(255)).astype(np.uint8)
        nc = (g.take(width * height // 2) >> np.uint32(24)) & np.uint32(7)
        out[n, width * height:] = (np.uint32(128) + nc).astype(np.uint8)
    return out
