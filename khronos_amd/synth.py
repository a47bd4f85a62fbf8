"""Synthetic RGB-D + label stream — ctypes wrapper over khronos_amd/synth/synth.cpp."""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkhr_synth.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run __graft_entry__.build()" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.synth_create.argtypes = [C.c_uint32, C.c_int, C.c_int]
        lib.synth_create.restype = C.c_void_p
        lib.synth_destroy.argtypes = [C.c_void_p]
        lib.synth_set_mover.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        lib.synth_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_void_p, C.c_double, C.c_float, C.c_float, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int]
        _lib = lib
    return _lib


def camera_pose(position, yaw):
    """world_T_sensor for an optical frame (x right, y down, z forward) looking along
    (cos yaw, sin yaw, 0) with world z up."""
    f = np.array([math.cos(yaw), math.sin(yaw), 0.0])
    r = np.array([f[1], -f[0], 0.0])
    d = np.array([0.0, 0.0, -1.0])
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = r, d, f, position
    return T


def circle_pose(t_sec, period=10.0, radius=1.5, height=1.5, yaw_offset=0.0):
    """camera on a circle of `radius` at `height`, yaw tangent (+ per-camera offset for rigs)."""
    th = 2.0 * math.pi * t_sec / period
    pos = np.array([radius * math.cos(th), radius * math.sin(th), height])
    return camera_pose(pos, th + math.pi / 2 + yaw_offset)


class SyntheticStream:
    """Deterministic scene + pinhole camera (fx = fy = W/2, 90 deg HFOV)."""

    def __init__(self, width, height, seed=1234, num_static=12, with_mover=True, max_depth=5.0, noise=0.0,
                 dt=0.1, period=10.0, threads=0):
        self.lib = _load()
        self.W, self.H = width, height
        self.fx = self.fy = width / 2.0
        self.cx, self.cy = width / 2.0, height / 2.0
        self.max_depth, self.noise, self.dt, self.period = max_depth, noise, dt, period
        self.seed, self.threads = seed, threads
        self.scene = self.lib.synth_create(seed, num_static, int(with_mover))

    def __del__(self):
        try:
            if self.scene:
                self.lib.synth_destroy(self.scene)
                self.scene = None
        except Exception:
            pass

    def set_mover(self, pos, vel, radius=0.3):
        p = np.asarray(pos, np.float32)
        v = np.asarray(vel, np.float32)
        self.lib.synth_set_mover(self.scene, p.ctypes.data, v.ctypes.data, radius)

    def stamp_ns(self, i):
        return int(round((1.0 + i * self.dt) * 1e9))

    def pose(self, i, yaw_offset=0.0):
        return circle_pose(i * self.dt, self.period, yaw_offset=yaw_offset)

    def render(self, i, pose=None, yaw_offset=0.0):
        T = np.ascontiguousarray(self.pose(i, yaw_offset) if pose is None else pose, dtype=np.float64)
        depth = np.empty((self.H, self.W), np.float32)
        rgb = np.empty((self.H, self.W, 3), np.uint8)
        label = np.empty((self.H, self.W), np.int32)
        self.lib.synth_render(self.scene, self.W, self.H, self.fx, self.fy, self.cx, self.cy, T.ctypes.data,
                              i * self.dt, self.max_depth, self.noise, self.seed + 7919 * i, depth.ctypes.data,
                              rgb.ctypes.data, label.ctypes.data, self.threads)
        return {"stamp": self.stamp_ns(i), "pose": T, "depth": depth, "rgb": rgb, "label": label}
