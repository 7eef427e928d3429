"""TEST INFRASTRUCTURE ONLY -- numpy/ctypes restatement of the transfer-function frame's state logic
and of the data probe, the checker for simian-spacemonkey_amd/host/TransferFunctions.{h,cpp}'s TFFrame /
probe_sample / place_brush.  PARITY UNPINNED by the reference (it holds no fixtures); every step cites
the lines it follows.  Widget painting itself is oracle.lev_rasterize (smk_prep.c).

  TFWidgetRen::drawFrame   TFWidgetRen1.cpp:194-242   clear paint / paint / drop / regenerate
  TFWidgetRen::init        TFWidgetRen1.cpp:648-656   the brush widget: alpha .7, ellipse
  TFWidgetRen::drawProbe   TFWidgetRen1.cpp:309-560   voxel cell under the probe, brush placement
  TFWidgetRen::triLerpV3   TFWidgetRen1.cpp:600-621
  LevWidget::rasterize     LevWidget.cpp:674-688      list order, scalar data modes force the 1-D style
"""
import copy

import numpy as np

import oracle as O

f32 = np.float32
NoBrush, EllipseBrush, AutoEllipseBrush, TriangleBrush, OneDBrush, AutoOneDBrush = range(6)
DM_V1, DM_VGH_V = 0, 11
KINDS = ["triangle", "ellipse", "1d", "default"]


class Widget:
    def __init__(self, kind="ellipse", b=(.5, 0), l=(.3, .7), r=(.7, .7), tw=-10.0, th=-10.0, hsl=(0, 1, .5), alpha=.5, be=1.0):
        self.kind, self.b, self.l, self.r, self.tw, self.th, self.hsl, self.alpha, self.be = kind, b, l, r, tw, th, hsl, alpha, be

    def paint(self, tex, dmode, faux):
        kind, b, l, r = self.kind, self.b, self.l, self.r
        w = O.lev_widget(kind, b=b, l=l, r=r, tw=self.tw, th=self.th, hsl=self.hsl, alpha=self.alpha, be=self.be, faux=faux)
        if dmode in (DM_V1, DM_VGH_V):        # LevWidget.cpp:677-682: verts' heights forced AFTER setPos, thresholds kept
            w.type = 2
            w.verts[0][1], w.verts[1][1], w.verts[2][1] = 0.0, 1.0, 1.0
        return O.lev_rasterize(w, tex)


class Frame:
    def __init__(self, sv, sg, sh, dmode, faux=True):
        self.shape = (sh, sg, sv, 4) if sh > 1 else (sg, sv, 4)
        self.dmode, self.faux = dmode, faux
        self.paintex = np.zeros(self.shape, np.uint8)
        self.widgets = []                       # newest first
        self.brush = Widget("ellipse", alpha=.7)
        self.brushon, self.brush_kind = False, NoBrush

    def clear_paint(self):
        self.paintex[:] = 0

    def paint(self):
        if self.brush_kind in (EllipseBrush, AutoEllipseBrush, OneDBrush, AutoOneDBrush):
            self.brush.paint(self.paintex, self.dmode, self.faux)
        elif self.brush_kind == TriangleBrush:
            self.widgets.insert(0, copy.deepcopy(self.brush))

    def drop(self):
        self.widgets.insert(0, copy.deepcopy(self.brush))

    def regenerate(self):
        tex = self.paintex.copy()
        for w in reversed(self.widgets):        # every widget rasterises its successors before itself
            w.paint(tex, self.dmode, self.faux)
        if self.brushon:
            self.brush.paint(tex, self.dmode, self.faux)
        return tex


def probe_sample(data, dmode, vpos):
    """data [z][y][x][ne] u8 -> (inside, cell, corners[8][3] f32, value[3] f32)"""
    sz, sy, sx, ne = data.shape
    v = [f32(x) for x in vpos]
    fpos = [v[0] * f32(sx), v[1] * f32(sy), v[2] * f32(sz)]
    px, py, pz = int(fpos[0]), int(fpos[1]), int(fpos[2])
    corners = np.zeros((8, 3), f32)
    inside = not (px < 1 or px > sx - 2 or py < 1 or py > sy - 2 or pz < 1 or pz > sz - 2)
    if not inside:
        return False, (px, py, pz), corners, np.zeros(3, f32)
    for i in range(2):
        for j in range(2):
            for k in range(2):
                d = data[pz + i, py + j, px + k]
                c = corners[i * 4 + j * 2 + k]
                c[0] = f32(float(d[0]) / 255.0)
                if dmode != DM_V1:
                    c[1] = f32(float(d[1]) / 255.0)
                    if dmode not in (1, 3, 4):          # not V1G / V2 / V2G: the third byte, /169 as the reference has it
                        c[2] = f32(float(d[2]) / 169.0)
    fx, fy, fz = (f32(fpos[a] - f32(int(fpos[a]))) for a in range(3))
    val = np.zeros(3, f32)
    for e in range(3):
        x1 = f32(corners[0][e] + f32(corners[1][e] - corners[0][e]) * fx)
        x2 = f32(corners[2][e] + f32(corners[3][e] - corners[2][e]) * fx)
        x3 = f32(corners[4][e] + f32(corners[5][e] - corners[4][e]) * fx)
        x4 = f32(corners[6][e] + f32(corners[7][e] - corners[6][e]) * fx)
        xy1 = f32(x1 + f32(x2 - x1) * fy)
        xy2 = f32(x3 + f32(x4 - x3) * fy)
        val[e] = f32(xy1 + f32(xy2 - xy1) * fz)
    return True, (px, py, pz), corners, val


def place_brush(frame, inside, corners, val, slider):
    """TFWidgetRen1.cpp:348-352 (outside) and :497-560: the brush widget follows the probe"""
    br, kind = frame.brush, frame.brush_kind
    slider = f32(slider)
    if not inside:
        br.b = br.l = br.r = (0.0, 0.0)
        br.tw = br.th = -10.0
        return
    v0, v1 = f32(val[0]), f32(val[1])
    if kind == EllipseBrush:
        bsz = f32((1.0 - float(slider)) / 4.0)
        br.l, br.r, br.b = (f32(v0 - bsz), f32(v1 + bsz)), (f32(v0 + bsz), f32(v1 + bsz)), (f32(v0 + bsz), f32(v1 - bsz))
        br.tw, br.th, br.kind = v0, v1, "ellipse"
    elif kind in (TriangleBrush, AutoEllipseBrush):
        maxx, minx = corners[:, 0].max(), corners[:, 0].min()
        maxy, miny = corners[:, 1].max(), corners[:, 1].min()
        bsz = f32((1.0 - float(slider)) * 2)
        w = f32(max(float(f32(maxx - minx)) / 2.0, .01))
        h = f32(max(float(f32(maxy - miny)) / 2.0, .01))
        br.l, br.r = (f32(v0 - f32(w * bsz)), f32(v1 + f32(h * bsz))), (f32(v0 + f32(w * bsz)), f32(v1 + f32(h * bsz)))
        if kind == TriangleBrush:
            br.b, br.tw, br.th, br.kind = (v0, 0.0), 0.0, f32(v1 - f32(h * bsz)), "triangle"
        else:
            br.b, br.tw, br.th, br.kind = (f32(v0 + f32(w * bsz)), f32(v1 - f32(h * bsz))), v0, v1, "ellipse"
    if frame.dmode in (DM_V1, DM_VGH_V) or kind in (OneDBrush, AutoOneDBrush):
        maxx, minx = corners[:, 0].max(), corners[:, 0].min()
        bsz = f32((1.0 - float(slider)) * 2)
        w = f32(max(float(f32(maxx - minx)) / 2.0, .01))
        l0, r0 = f32(v0 - f32(w * bsz)), f32(v0 + f32(w * bsz))
        if kind == OneDBrush:
            l0, r0 = f32(float(v0) + (1.0 - float(slider)) / 4.0), f32(float(v0) - (1.0 - float(slider)) / 4.0)
        br.l, br.r, br.b, br.tw, br.th, br.kind = (l0, 1.0), (r0, 1.0), (v0, 0.0), v0, v1, "1d"
    br.b, br.l, br.r = tuple(float(x) for x in br.b), tuple(float(x) for x in br.l), tuple(float(x) for x in br.r)
    br.tw, br.th = float(br.tw), float(br.th)
