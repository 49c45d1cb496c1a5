#!/usr/bin/env python3
"""How long the host's Horner pass over a call's bit-positioned points takes (curdle_window_combine, the
serial tail of every synchronous MSM): the verifier's 1,368-pair MSM (13 windows x 11 points) and N = 2^20
(8 windows x 14 points), on this machine's CPU.  No GPU needed."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import curdlemsm as cm
import bls12381_ref as o
import coracle as co

lib = cm._lib
k, q = o.Rand(7).get_frs(2)
Rm = (1 << 384) % o.P
onel = np.array([(Rm >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], dtype=np.uint64)
add = lib.curdle_host_add
add.argtypes = [C.c_void_p, C.c_void_p]
wc = lib.curdle_window_combine
wc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]


def case(name, nw, c, rel):
    n = nw * len(rel)
    pts = co.points_walk(k, q, n)
    xy = np.zeros((n, 24), dtype=np.uint64)
    xy[:, :12] = pts
    xy[:, 12:18] = onel
    xy[:, 18:24] = onel
    for i in range(n):                      # general ZZ: every point the sum of two
        a = xy[i].copy()
        add(a.ctypes.data, xy[(i + 1) % n].ctypes.data)
        xy[i] = a
    pos = [c * w + r for w in range(nw) for r in rel]
    dbls = np.zeros(n, dtype=np.int32)
    prev = 0
    for i, pp in enumerate(pos):
        dbls[i] = pp - prev
        prev = pp
    out = np.zeros(18, dtype=np.uint64)
    for _ in range(200):
        wc(xy.ctypes.data, n, dbls.ctypes.data, out.ctypes.data)
    t = time.perf_counter()
    for _ in range(2000):
        wc(xy.ctypes.data, n, dbls.ctypes.data, out.ctypes.data)
    us = (time.perf_counter() - t) / 2000 * 1e6
    print("%-44s %3d points, %3d doublings: %6.1f us" % (name, n, int(dbls.sum()), us))


case("verifier's MSM (c = 10, 13 windows x 11)", 13, 10, [0] + list(range(0, 10)))
case("N = 2^20 (c = 16, 8 windows x 14)", 8, 16, [0] + [3 + i for i in range(13)])
case("window sums only (c = 16, 8 x 1)", 8, 16, [0])
