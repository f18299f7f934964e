"""On-disk formats around the hot path: PFM height maps and the 170-line `.rpc` text files of WHU-TLC
(/root/reference/dataset/data_io.py:17-92).  Host-side numpy, same function names and return values as the reference.
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np


def load_pfm(fname):
    """PFM ('Pf' grey / 'PF' colour, scale sign = endianness) -> float32 array, top row first (data_io.py:17-44)."""
    with open(fname, "rb") as f:
        header = f.readline().decode("latin-1").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("latin-1"))
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = (int(v) for v in dims.groups())
        scale = float(f.readline().decode("latin-1").rstrip())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flip(data.reshape(shape), 0).copy()          # writable and contiguous, like the reference's np.fromfile + flipud


def save_pfm(file, image, scale=1):
    """float32 (H,W) / (H,W,1) / (H,W,3) -> PFM, bottom row first, negative scale for little-endian (data_io.py:47-75)."""
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    image = np.flipud(image)
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(file, "wb") as f:
        f.write(("PF\n" if color else "Pf\n").encode("utf8"))
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode("utf8"))
        f.write(("%f\n" % (-scale if little else scale)).encode("utf8"))
        f.write(np.ascontiguousarray(image).tobytes())


def load_rpc_as_array(filepath):
    """`.rpc` text (one 'NAME value' pair per line, 170 lines in the order of tools/RPCCore.py:8-28) ->
    (float64[170], h_max, h_min) with h = HEIGHT_OFF +/- HEIGHT_SCALE (data_io.py:78-92)."""
    if not os.path.exists(filepath):
        raise Exception("RPC not found! Can not find " + filepath + " in the file system!")
    with open(filepath, "r") as f:
        lines = f.read().splitlines()
    data = np.array([t.split(" ")[1] for t in lines], dtype=np.float64)
    return data, data[4] + data[9], data[4] - data[9]


# The 20 RPC monomials in the order of tools/RPCCore.py:8-28 as index triples over (1, L, P, H): coefficient n multiplies
# v[i] * v[j] * v[k].  The quaternary-cubic ("QC") form spreads it evenly over the permutations of its triple in a symmetric
# (4,4,4) tensor, so that  poly = sum_ijk T[i,j,k] v_i v_j v_k  (data_io.py:95-120).
_QC_TRIPLES = [(0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 1, 2), (0, 1, 3), (0, 2, 3), (0, 1, 1), (0, 2, 2), (0, 3, 3),
               (1, 2, 3), (1, 1, 1), (1, 2, 2), (1, 3, 3), (1, 1, 2), (2, 2, 2), (2, 3, 3), (1, 1, 3), (2, 2, 3), (3, 3, 3)]
_QC_SCALARS = ["line_off", "samp_off", "lat_off", "lon_off", "height_off", "line_scale", "samp_scale", "lat_scale", "lon_scale",
               "height_scale"]
_QC_TENSORS = ["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]


def to_tensor(data):
    """20 cubic coefficients -> the symmetric (4,4,4) float64 tensor of the `use_qc` operators: every permutation of a monomial's
    index triple holds coefficient / (number of distinct permutations) -- 1, 3 or 6 (data_io.py:95-120)."""
    import itertools
    data = np.asarray(data, np.float64)
    assert data.shape == (20,)
    out = np.zeros((4, 4, 4), np.float64)
    for c, t in zip(data, _QC_TRIPLES):
        perms = set(itertools.permutations(t))
        for q in perms:
            out[q] = c / float(len(perms))
    return out


def load_rpc_as_qc_tensor(filepath):
    """`.rpc` text -> the dict the `use_qc=True` networks consume (data_io.py:123-150): ten float64 scalars `line_off` ...
    `height_scale` and eight `<line|samp|lat|lon>_<num|den>_tensor` (4,4,4) arrays."""
    data = load_rpc_as_array(filepath)[0]
    rpc = {k: data[i] for i, k in enumerate(_QC_SCALARS)}
    for n, k in enumerate(_QC_TENSORS):
        rpc[k + "_tensor"] = to_tensor(data[10 + 20 * n: 30 + 20 * n])
    return rpc


_RPC_NAMES = (["LINE_OFF", "SAMP_OFF", "LAT_OFF", "LONG_OFF", "HEIGHT_OFF", "LINE_SCALE", "SAMP_SCALE", "LAT_SCALE",
               "LONG_SCALE", "HEIGHT_SCALE"]
              + ["%s_%d" % (n, i + 1) for n in ("LINE_NUM_COEFF", "LINE_DEN_COEFF", "SAMP_NUM_COEFF", "SAMP_DEN_COEFF",
                                               "LAT_NUM_COEFF", "LAT_DEN_COEFF", "LONG_NUM_COEFF", "LONG_DEN_COEFF")
                 for i in range(20)])


def save_rpc(filepath, rpc170):
    """Inverse of load_rpc_as_array (the reference has no writer): 170 'NAME value' lines, 17 significant digits."""
    rpc170 = np.asarray(rpc170, np.float64).reshape(-1)
    if rpc170.size != 170:
        raise ValueError("an RPC vector holds 170 values")
    with open(filepath, "w") as f:
        for n, v in zip(_RPC_NAMES, rpc170):
            f.write("%s %.17g\n" % (n, v))
