"""TEST INFRASTRUCTURE (the checker, never the product path): `lidar_occupancy` EXACTLY as the reference computes it.

The reference's `OccupancyMapObs.step` (dreamer/wrappers.py:396-406) is the one function of the hot path whose arithmetic IS in
the reference tree - through two third-party libraries:

    pr, pc = map.to_pixel(pose)                                        # :396   row = int(H - (y - oy) / res), col = int((x - ox) / res)
    track_map = map._map[pr - 110:pr + 110, pc - 110:pc + 110]        # :398   220 x 220 crop, north-up, 1 = drivable
    track_map = ndimage.rotate(track_map.astype(uint8), rad2deg(2 pi - yaw))      # :401   cubic spline, reshape, constant 0
    cropped = track_map[cr - 100:cr + 100, cc - 100:cc + 100]         # :402-404 centre 200 x 200
    cropped = Image.fromarray(cropped).resize((64, 64))                # :405   bicubic with antialiasing (support 6.25 px)

Two statements of it live here:

* `render_patch_reference(track, poses)` - those very library calls (scipy.ndimage.rotate, PIL.Image.resize: third-party code,
  not reference source) on the full-frame drivable grid.  Pinned 100 % to the 379 G6 patches the reference's own wrapper
  produced (tests/test_golden_patch.py); needs scipy + Pillow, so it exists only on the CPU side.
* `render_patch_exact(track, poses)` - the SPEC of obs_type `lidar_occupancy_reference`: the same pipeline restated down to the
  IEEE binary64 operation, with no library behind it, which the C oracle (`oc_patch_exact_range`) and the HIP kernel
  (`rc_patch_exact_kernel`) follow operation by operation:
    1. crop: 220 x 220 cells of the drivable grid around (pr, pc), 0 outside the grid (the reference raises or wraps there);
    2. spline prefilter of scipy >= 1.6 (ndimage/src/ni_splines.c: mode 'constant' filters as 'mirror'): per line, along axis 0
       then axis 1, c *= (1 - 1/z)(1 - z); c[0] = (c[0] + z^(n-1) c[n-1] + sum_i z^i (c[i] + z^(n-1) c[n-1-i])) / (1 - z^(2n-2));
       c[i] += z c[i-1]; c[n-1] = (z c[n-2] + c[n-1]) z / (z z - 1); c[i] = z (c[i+1] - c[i]); z = -0.2679491924311227 (the
       correctly rounded sqrt(3) - 2 the library carries as a literal; sqrt(3.0) - 2.0 evaluated in binary64 is 2 ulp off);
    3. rotation matrix [[c, s], [-s, c]] with c, s = cos, sin of the angle in DEGREES a = (2 pi - yaw) (180 / pi): exact reduction
       mod 45 degrees, Taylor polynomials on |z| <= pi / 4 (this build's own; scipy's cephes cosdg / sindg are not restated - the
       two agree to ~1e-16, which moves a tap by < 1e-13 cell);
       output shape S = int(ptp(bounds) + 0.5) per axis, offset = 109.5 - M (S - 1) / 2, centre window rows S0 // 2 - 100 ...;
    4. per output pixel: input coordinate cc_h = ((0 + o0 M[h][0]) + o1 M[h][1]) + offset_h; outside [0, 219] on either axis ->
       0; else 4 x 4 taps from floor(cc) - 1 (mirrored at the edges), weights y = cc - floor(cc), z = 1 - y:
       w1 = ((y y)(y - 2) 3 + 4) / 6, w2 = ((z - 2)(z z) 3 + 4) / 6, w0 = ((z z) z) / 6, w3 = ((1 - w0) - w1) - w2;
       t = sum over (i, j) in row-major order of (coef w0_i) w1_j; value = uint8(t + 0.5) if t > 0 else 0, clamped to 255;
    5. Pillow's 8-bit resize, horizontal pass then vertical: integer coefficients round(k 2^22) of the normalised bicubic
       (a = -0.5) kernel stretched by 3.125, accumulator 2^21 + sum(pixel k) >> 22 clamped to [0, 255] - integers throughout.
  Steps 2 and 4 equal scipy's float64 output BIT FOR BIT when given the same matrix and offset; steps 3's cos / sin and the BLAS
  behind scipy's `rot_matrix @ ...` are where the library's last bit may differ, so what is asserted against the library is the
  uint8 image: identical on all 379 G6 patches and on 10^4 random poses per track (tools/analysis/patch_reference_divergence.py).
"""
from __future__ import annotations

import numpy as np

Z_POLE = -0.2679491924311227                 # scipy ndimage/src/ni_splines.c get_filter_poles(3): sqrt(3) - 2, correctly rounded
Z_POW_219 = -5.539710763905135e-126          # pow(Z_POLE, 220 - 1): the crop is always 220 cells wide
CROP, WIN, OUT = 220, 200, 64                # dreamer/wrappers.py:374,378,398-405
PI180 = 1.74532925199432957692e-2
PRECISION_BITS = 22                          # Pillow Resample.c: 32 - 8 - 2


# ------------------------------------------------------------------------------------------------ geometry of the full frame
def frame_of(track):
    """(fh, ox, oy, r_top, c0): the source image's height, the world position of its lower left corner, and how a NORTH-UP
    full-frame pixel (R, C) maps to a cell of the track's own south-up cropped grid: gy = r_top - R, gx = C - c0."""
    r0, c0, fh, _fw = track.crop
    res = float(track.resolution)
    ox = float(track.origin[0]) - c0 * res
    oy = float(track.origin[1]) - (fh - (r0 + track.height)) * res
    return int(fh), ox, oy, int(r0 + track.height - 1), int(c0)


def to_pixel(track, x, y):
    """racecar_gym's GridMap.to_pixel on the full frame (compat/racecar_gym/core/gridmaps.py): float64,
    truncation towards zero."""
    fh, ox, oy, _, _ = frame_of(track)
    res = float(track.resolution)
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return np.trunc(fh - (y - oy) / res).astype(np.int64), np.trunc((x - ox) / res).astype(np.int64)


def crops(track, pr, pc):
    """uint8 [n, 220, 220]: rows pr - 110 .. pr + 109, columns pc - 110 .. pc + 109 of the north-up full frame, 0 outside the grid."""
    _, _, _, r_top, c0 = frame_of(track)
    drv = np.asarray(track.drivable, bool).copy()
    drv[0, :] = drv[-1, :] = drv[:, 0] = drv[:, -1] = False               # the env's spec: the outermost ring of cells is not drivable
    h, w = drv.shape
    r = np.arange(-CROP // 2, CROP // 2)
    gy = r_top - (np.asarray(pr)[:, None] + r[None, :])                   # [n, 220]
    gx = (np.asarray(pc)[:, None] + r[None, :]) - c0
    iny, inx = (gy >= 0) & (gy < h), (gx >= 0) & (gx < w)
    out = drv[np.clip(gy, 0, h - 1)[:, :, None], np.clip(gx, 0, w - 1)[:, None, :]]
    return (out & iny[:, :, None] & inx[:, None, :]).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ the library path (checker)
def render_patch_reference(track, poses):
    """The reference's own call sequence on third-party libraries (needs scipy and Pillow).  poses float64 [n, 3] = x, y, yaw.
    uint8 [n, 64, 64], values 0 / 1."""
    from PIL import Image
    from scipy import ndimage
    poses = np.asarray(poses, np.float64).reshape(-1, 3)
    pr, pc = to_pixel(track, poses[:, 0], poses[:, 1])
    out = np.zeros((len(poses), OUT, OUT), np.uint8)
    for k, crop in enumerate(crops(track, pr, pc)):
        rot = ndimage.rotate(crop, np.rad2deg(2 * np.pi - poses[k, 2]))
        cr, cc = rot.shape[0] // 2, rot.shape[1] // 2
        out[k] = np.array(Image.fromarray(rot[cr - WIN // 2:cr + WIN // 2, cc - WIN // 2:cc + WIN // 2]).resize(size=(OUT, OUT)))
    return out


# ------------------------------------------------------------------------------------------------ the restatement (the spec)
def _filter_lines(c):
    """Cubic-spline prefilter along the LAST axis of c [..., 220] (float64, in place)."""
    z, n = Z_POLE, c.shape[-1]
    c *= (1.0 - 1.0 / z) * (1.0 - z)
    zn = Z_POW_219 if n == CROP else z ** (n - 1)
    c0 = c[..., 0] + zn * c[..., n - 1]
    zi = z
    for i in range(1, n - 1):
        c0 = c0 + zi * (c[..., i] + zn * c[..., n - 1 - i])
        zi *= z
    c[..., 0] = c0 / (1.0 - zn * zn)
    for i in range(1, n):
        c[..., i] += z * c[..., i - 1]
    c[..., n - 1] = (z * c[..., n - 2] + c[..., n - 1]) * z / (z * z - 1.0)
    for i in range(n - 2, -1, -1):
        c[..., i] = z * (c[..., i + 1] - c[..., i])
    return c


def spline_coefficients(crop):
    """uint8 [n, 220, 220] -> float64 [n, 220, 220]: scipy.ndimage.spline_filter(order 3), axis 0 then axis 1."""
    c = crop.astype(np.float64)
    c = np.swapaxes(_filter_lines(np.ascontiguousarray(np.swapaxes(c, 1, 2))), 1, 2)
    return _filter_lines(np.ascontiguousarray(c))


def sincos_degrees(deg):
    """(cos, sin) of a non-negative angle in degrees: octant by an exact reduction mod 45, Taylor polynomials on |z| <= pi / 4."""
    x = np.asarray(deg, np.float64)
    y = np.floor(x / 45.0)
    j = (y - 8.0 * np.floor(y / 8.0)).astype(np.int64)
    odd = (j & 1) == 1
    y = np.where(odd, y + 1.0, y)
    j = np.where(odd, j + 1, j) & 7
    z = (x - y * 45.0) * PI180
    zz = z * z
    sp = z + z * (zz * (-1.0 / 6.0 + zz * (1.0 / 120.0 + zz * (-1.0 / 5040.0 + zz * (1.0 / 362880.0 + zz * (-1.0 / 39916800.0 + zz * (
        1.0 / 6227020800.0 + zz * (-1.0 / 1307674368000.0))))))))
    cp = 1.0 - zz * (0.5 - zz * (1.0 / 24.0 - zz * (1.0 / 720.0 - zz * (1.0 / 40320.0 - zz * (1.0 / 3628800.0 - zz * (1.0 / 479001600.0 - zz * (
        1.0 / 87178291200.0 - zz * (1.0 / 20922789888000.0))))))))
    cos = np.where(j == 0, cp, np.where(j == 2, -sp, np.where(j == 4, -cp, sp)))
    sin = np.where(j == 0, sp, np.where(j == 2, cp, np.where(j == 4, -sp, -cp)))
    return cos, sin


def rotation(yaw):
    """Per pose (yaw float64 [n]): c, s, output shape S0, S1 and offset off0, off1 of scipy.ndimage.rotate(reshape=True) on a
    220 x 220 input."""
    deg = (2.0 * np.pi - np.asarray(yaw, np.float64)) * (180.0 / np.pi)
    c, s = sincos_degrees(deg)
    n = float(CROP)
    b0 = np.stack([c * 0.0 + s * 0.0, c * 0.0 + s * n, c * n + s * 0.0, c * n + s * n])
    b1 = np.stack([-s * 0.0 + c * 0.0, -s * 0.0 + c * n, -s * n + c * 0.0, -s * n + c * n])
    s0 = ((b0.max(0) - b0.min(0)) + 0.5).astype(np.int64)
    s1 = ((b1.max(0) - b1.min(0)) + 0.5).astype(np.int64)
    h0, h1 = (s0 - 1) / 2, (s1 - 1) / 2
    off0 = (CROP - 1) / 2 - (c * h0 + s * h1)
    off1 = (CROP - 1) / 2 - (-s * h0 + c * h1)
    return c, s, s0, s1, off0, off1


def _weights(cc):
    y = cc - np.floor(cc)
    z = 1.0 - y
    w1 = ((y * y) * (y - 2.0) * 3.0 + 4.0) / 6.0
    w2 = ((z - 2.0) * (z * z) * 3.0 + 4.0) / 6.0
    w0 = ((z * z) * z) / 6.0
    return w0, w1, w2, ((1.0 - w0) - w1) - w2


def _mirror(idx):
    idx = np.where(idx < 0, -idx, idx)
    return np.where(idx >= CROP, 2 * CROP - 2 - idx, idx)


def rotated_window(coef, yaw):
    """float64 spline coefficients [n, 220, 220] + yaw [n] -> uint8 [n, 200, 200]: the centre window of the rotated image."""
    n = len(coef)
    c, s, s0, s1, off0, off1 = rotation(yaw)
    i = np.arange(WIN, dtype=np.float64)
    o0 = ((s0 // 2 - WIN // 2).astype(np.float64)[:, None] + i[None, :])[:, :, None]                # [n, 200, 1]
    o1 = ((s1 // 2 - WIN // 2).astype(np.float64)[:, None] + i[None, :])[:, None, :]                # [n, 1, 200]
    c_, s_ = c[:, None, None], s[:, None, None]
    cc0 = ((0.0 + o0 * c_) + o1 * s_) + off0[:, None, None]
    cc1 = ((0.0 + o0 * (-s_)) + o1 * c_) + off1[:, None, None]
    const = (cc0 < 0) | (cc0 > CROP - 1) | (cc1 < 0) | (cc1 > CROP - 1)
    cc0, cc1 = np.where(const, 0.0, cc0), np.where(const, 0.0, cc1)
    st0, st1 = np.floor(cc0).astype(np.int64) - 1, np.floor(cc1).astype(np.int64) - 1
    w0, w1 = _weights(cc0), _weights(cc1)
    k = np.arange(n)[:, None, None]
    t = np.zeros((n, WIN, WIN))
    for a in range(4):
        r = _mirror(st0 + a)
        for b in range(4):
            t = t + (coef[k, r, _mirror(st1 + b)] * w0[a]) * w1[b]
    t = np.where(const, 0.0, t)
    t = np.where(t > 0, t + 0.5, 0.0)
    return np.minimum(t, 255.0).astype(np.uint8)


def _bicubic(x):
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resize_coefficients(in_size=WIN, out_size=OUT):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the bicubic filter: int64
    [out, ksize] coefficients x 2^22 and [out, 2] (first input pixel, number of pixels)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    kk, bounds = np.zeros((out_size, ksize), np.int64), np.zeros((out_size, 2), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x, w in enumerate(k):
            w = w / ww if ww != 0.0 else w
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


_KK = None


def resize(img):
    """uint8 [n, 200, 200] -> uint8 [n, 64, 64]: Pillow's 8-bit bicubic resize, horizontal pass then vertical, in integers."""
    global _KK
    if _KK is None:
        _KK = resize_coefficients()
    kk, bounds = _KK
    im = img.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.zeros((len(im), WIN, OUT), np.int64)
    for xx in range(OUT):
        x0, xm = bounds[xx]
        tmp[:, :, xx] = np.clip((half + (im[:, :, x0:x0 + xm] * kk[xx, :xm]).sum(-1)) >> PRECISION_BITS, 0, 255)
    out = np.zeros((len(im), OUT, OUT), np.int64)
    for yy in range(OUT):
        y0, ym = bounds[yy]
        out[:, yy, :] = np.clip((half + (tmp[:, y0:y0 + ym, :] * kk[yy, :ym, None]).sum(1)) >> PRECISION_BITS, 0, 255)
    return out.astype(np.uint8)


def render_patch_exact(track, poses, chunk=8):
    """The spec of obs_type `lidar_occupancy_reference`.  poses [n, 3] = x, y, yaw (the env's float32 state widened to
    float64).  uint8 [n, 64, 64]."""
    poses = np.asarray(poses, np.float64).reshape(-1, 3)
    out = np.zeros((len(poses), OUT, OUT), np.uint8)
    for a in range(0, len(poses), chunk):
        p = poses[a:a + chunk]
        pr, pc = to_pixel(track, p[:, 0], p[:, 1])
        out[a:a + chunk] = resize(rotated_window(spline_coefficients(crops(track, pr, pc)), p[:, 2]))
    return out
