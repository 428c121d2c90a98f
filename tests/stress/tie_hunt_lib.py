"""Voxel-averaged level of a cloud (rs_pointcloud.h:905-975)."""
import numpy as np


def level(pos, nor, voxel):
    mn = pos.min(0)
    key = np.floor((pos - mn) / np.float32(voxel)).astype(np.int64)
    key = (key[:, 2] * 100003 + key[:, 1]) * 100003 + key[:, 0]
    order = np.argsort(key, kind="stable"); key = key[order]
    start = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    cnt = np.diff(np.r_[start, len(key)]).astype(np.float32)[:, None]
    p = np.add.reduceat(pos[order], start, axis=0) / cnt
    n = np.add.reduceat(nor[order], start, axis=0) / cnt
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)
    return p.astype(np.float32), n.astype(np.float32)
