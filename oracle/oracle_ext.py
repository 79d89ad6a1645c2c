"""CPU ORACLE façade -- test infrastructure, NOT product code.

Exposes the nine callables of the reference's ``pointnet2._ext`` pybind module
(/root/reference/pointnet2/_ext_src/src/bindings.cpp:11-24) on top of
``oracle/libeda_oracle.so`` for CPU torch tensors.  Used by

* ``tests/`` as the parity checker for the HIP path,
* ``tools/gen_golden.py`` (build container only) as the ``pointnet2._ext``
  stand-in when the reference's Python layers are imported to make fixtures,
* ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()``.

Nothing under ``eda_amd/`` imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libeda_oracle.so")

_c_f = ctypes.POINTER(ctypes.c_float)
_c_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "eda_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        i, f = ctypes.c_int, ctypes.c_float
        sig = {
            "eda_oracle_furthest_point_sampling": [i, i, i, _c_f, _c_f, _c_i],
            "eda_oracle_furthest_point_sampling_mt": [i, i, i, _c_f, _c_f, _c_i],
            "eda_oracle_gather_points": [i, i, i, i, _c_f, _c_i, _c_f],
            "eda_oracle_gather_points_grad": [i, i, i, i, _c_f, _c_i, _c_f],
            "eda_oracle_ball_query": [i, i, i, f, i, _c_f, _c_f, _c_i],
            "eda_oracle_ball_query_mt": [i, i, i, f, i, _c_f, _c_f, _c_i],
            "eda_oracle_group_points": [i, i, i, i, i, _c_f, _c_i, _c_f],
            "eda_oracle_group_points_mt": [i, i, i, i, i, _c_f, _c_i, _c_f],
            "eda_oracle_group_points_grad": [i, i, i, i, i, _c_f, _c_i, _c_f],
            "eda_oracle_three_nn": [i, i, i, _c_f, _c_f, _c_f, _c_i],
            "eda_oracle_three_interpolate": [i, i, i, i, _c_f, _c_i, _c_f, _c_f],
            "eda_oracle_three_interpolate_grad": [i, i, i, i, _c_f, _c_i, _c_f, _c_f],
            "eda_oracle_set_fma_mode": [i],
            "eda_oracle_set_threads": [i],
            "eda_oracle_opt_n_threads": [i],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = None
        for name in ("eda_oracle_get_fma_mode", "eda_oracle_get_threads", "eda_oracle_opt_n_threads"):
            getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


def set_fma_mode(mode):
    lib().eda_oracle_set_fma_mode(int(mode))


def set_threads(n):
    lib().eda_oracle_set_threads(int(n))


def opt_n_threads(w):
    return lib().eda_oracle_opt_n_threads(int(w))


def _fp(t):
    return ctypes.cast(t.data_ptr(), _c_f)


def _ip(t):
    return ctypes.cast(t.data_ptr(), _c_i)


# The reference's argument checks (include/utils.h:10-30), same messages.
def _chk_contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _chk_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def _chk_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _chk_cpu(t, name):
    if t.device.type != "cpu":
        raise RuntimeError(f"{name}: the oracle only takes CPU tensors")


def furthest_point_sampling(points, nsamples, mt=False):
    _chk_contig(points, "points"); _chk_float(points, "points"); _chk_cpu(points, "points")
    b, n = points.shape[0], points.shape[1]
    out = torch.zeros((b, nsamples), dtype=torch.int32)
    tmp = torch.empty((b, n), dtype=torch.float32)
    fn = lib().eda_oracle_furthest_point_sampling_mt if mt else lib().eda_oracle_furthest_point_sampling
    fn(b, n, int(nsamples), _fp(points), _fp(tmp), _ip(out))
    return out


def gather_points(points, idx):
    _chk_contig(points, "points"); _chk_contig(idx, "idx")
    _chk_float(points, "points"); _chk_int(idx, "idx"); _chk_cpu(points, "points")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.zeros((b, c, m), dtype=torch.float32)
    lib().eda_oracle_gather_points(b, c, n, m, _fp(points), _ip(idx), _fp(out))
    return out


def gather_points_grad(grad_out, idx, n):
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx"); _chk_cpu(grad_out, "grad_out")
    b, c, m = grad_out.shape
    out = torch.zeros((b, c, int(n)), dtype=torch.float32)
    lib().eda_oracle_gather_points_grad(b, c, int(n), m, _fp(grad_out), _ip(idx), _fp(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample, mt=False):
    _chk_contig(new_xyz, "new_xyz"); _chk_contig(xyz, "xyz")
    _chk_float(new_xyz, "new_xyz"); _chk_float(xyz, "xyz"); _chk_cpu(xyz, "xyz")
    b, n = xyz.shape[0], xyz.shape[1]
    m = new_xyz.shape[1]
    idx = torch.zeros((new_xyz.shape[0], m, int(nsample)), dtype=torch.int32)
    fn = lib().eda_oracle_ball_query_mt if mt else lib().eda_oracle_ball_query
    fn(b, n, m, float(radius), int(nsample), _fp(new_xyz), _fp(xyz), _ip(idx))
    return idx


def group_points(points, idx, mt=False):
    _chk_contig(points, "points"); _chk_contig(idx, "idx")
    _chk_float(points, "points"); _chk_int(idx, "idx"); _chk_cpu(points, "points")
    b, c, n = points.shape
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = torch.zeros((b, c, npoints, nsample), dtype=torch.float32)
    fn = lib().eda_oracle_group_points_mt if mt else lib().eda_oracle_group_points
    fn(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out))
    return out


def group_points_grad(grad_out, idx, n):
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx"); _chk_cpu(grad_out, "grad_out")
    b, c = grad_out.shape[0], grad_out.shape[1]
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = torch.zeros((b, c, int(n)), dtype=torch.float32)
    lib().eda_oracle_group_points_grad(b, c, int(n), npoints, nsample, _fp(grad_out), _ip(idx), _fp(out))
    return out


def three_nn(unknowns, knows):
    _chk_contig(unknowns, "unknowns"); _chk_contig(knows, "knows")
    _chk_float(unknowns, "unknowns"); _chk_float(knows, "knows"); _chk_cpu(knows, "knows")
    b, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    idx = torch.zeros((b, n, 3), dtype=torch.int32)
    dist2 = torch.zeros((b, n, 3), dtype=torch.float32)
    lib().eda_oracle_three_nn(b, n, m, _fp(unknowns), _fp(knows), _fp(dist2), _ip(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    _chk_contig(points, "points"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(points, "points"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    _chk_cpu(points, "points")
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros((b, c, n), dtype=torch.float32)
    lib().eda_oracle_three_interpolate(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    _chk_cpu(grad_out, "grad_out")
    b, c, n = grad_out.shape
    out = torch.zeros((b, c, int(m)), dtype=torch.float32)
    lib().eda_oracle_three_interpolate_grad(b, c, n, int(m), _fp(grad_out), _ip(idx), _fp(weight), _fp(out))
    return out
