"""TEST INFRASTRUCTURE (never shipped, never linked by the product): a numpy restatement of the empty-space flags
of simian-spacemonkey_amd/csrc/smk_bricks.hip, rule for rule, so that the GPU's flags can be compared byte for byte
(tests/test_gpu_bricks.py) and the rule's soundness checked on the CPU (tests/test_bricks_kat.py).

The reference has no such structure -- it draws every slice of every brick and lets the blend unit discard what the
table made transparent (VolumeRenderer.cpp:507-741, NV20VolRen3D.cpp:852-1083) -- so there is nothing of the
reference's to pin this against; what IS checked is that a clear flag implies an exactly transparent sample.

    brick           8x8x8 cells; its value range = min/max of channels 0 and 1 over the (8+1)^3 voxels its cells touch
    occupancy bit   (t, s): some texel of the 2x2 quad based at (s, t) has alpha != 0  (smk_api.hip refresh_tf2d)
    flag            any occupancy bit in [base(vmin)-1, base(vmax)+1] x [base(gmin)-1, base(gmax)+1],
                    base(c) = min(int(clamp(fma(c, size, -0.5), 0, size-1)), size-2)     (smk_lin_clamp)
"""
import numpy as np

BRICK = 8


def occupancy(alpha):
    """alpha [sg][sv] uint8 (the EFFECTIVE table's alpha) -> bool [sg][sv]: the quad based at (s, t) is not all zero"""
    a = alpha != 0
    q = a.copy()
    q[:-1, :] |= a[1:, :]
    q[:, :-1] |= a[:, 1:]
    q[:-1, :-1] |= a[1:, 1:]
    return q


def fold_occupancy(alpha3):
    """dense 3-D table alpha [sh][sg][sv] -> the (v, g) occupancy folded over the sheets (smk_set_tf3d)"""
    q = np.zeros(alpha3.shape[1:], bool)
    for h in range(alpha3.shape[0]):
        q |= occupancy(alpha3[h])
    return q


def base_texel(c, size):
    """smk_lin_clamp's base index of channel value c (float32) in a table of `size` texels"""
    x = (c.astype(np.float64) * float(size) - 0.5).astype(np.float32)     # one rounding: the fma
    x = np.minimum(np.maximum(x, np.float32(0.0)), np.float32(size - 1))
    return np.minimum(x.astype(np.int64), max(size - 2, 0))


def brick_ranges(v, g):
    """v, g [D2][D1][D0] float32 (byte voxels: byte * float32(1/255)) -> (vmin, vmax, gmin, gmax) per brick [nb2][nb1][nb0]"""
    D = v.shape
    nb = [(d - 1) // BRICK + 1 for d in D]
    out = [np.zeros(nb, np.float32) for _ in range(4)]
    for bz in range(nb[0]):
        for by in range(nb[1]):
            for bx in range(nb[2]):
                sl = (slice(bz * BRICK, bz * BRICK + BRICK + 1), slice(by * BRICK, by * BRICK + BRICK + 1),
                      slice(bx * BRICK, bx * BRICK + BRICK + 1))
                out[0][bz, by, bx] = v[sl].min()
                out[1][bz, by, bx] = v[sl].max()
                out[2][bz, by, bx] = g[sl].min()
                out[3][bz, by, bx] = g[sl].max()
    return out


def brick_flags(v, g, occ):
    """the flags: uint8 [nb2][nb1][nb0]"""
    sg, sv = occ.shape
    vmin, vmax, gmin, gmax = brick_ranges(v, g)
    s_lo = np.maximum(base_texel(vmin, sv) - 1, 0)
    s_hi = np.minimum(base_texel(vmax, sv) + 1, sv - 1)
    t_lo = np.maximum(base_texel(gmin, sg) - 1, 0)
    t_hi = np.minimum(base_texel(gmax, sg) + 1, sg - 1)
    sat = np.zeros((sg + 1, sv + 1), np.int64)
    sat[1:, 1:] = np.cumsum(np.cumsum(occ.astype(np.int64), 0), 1)
    n = sat[t_hi + 1, s_hi + 1] - sat[t_lo, s_hi + 1] - sat[t_hi + 1, s_lo] + sat[t_lo, s_lo]
    return (n != 0).astype(np.uint8)
