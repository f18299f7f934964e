"""The caller side of the hot path that is still needed to run a WHU-TLC tile end to end: PNG views, the sample list and the
per-tile sample assembler (/root/reference/dataset/satmvsdataset.py:36-160, dataset/gen_list.py:6-84,
dataset/data_io.py:154-166, dataset/preprocess.py:150-160).  Host-side numpy + PIL, same names, dictionary keys, dtypes and
values as the reference (pinned by tests/golden/dataset.npz: the reference's own MVSDataset run on tests/golden/scene/).

A scene folder holds  image/<view>/<tile>.png,  rpc/<view>/<tile>.rpc,  height/<view>/<tile>.pfm  for views 0..V-1.
A sample is what networks/casred.py consumes: "imgs" (V,3,H,W) float32, each view normalised to zero mean / unit variance per
channel; "cam_para" {"stage1": rpc at 1/4 resolution, "stage2": 1/2, "stage3": full} (V,170) float64; "depth_values"
[h_min, h_max] of the ref view; and, outside "pred" mode, the ground-truth height map and its validity mask at the three
scales (nearest-neighbour decimation).  With `use_qc=True` "cam_para" holds, per stage, a list of V dictionaries in the
quaternary-cubic tensor form of the RPCs (satmvsdataset.py:166-296, data_io.py:123-150) -- what the `use_qc=True` networks index.
"""
from __future__ import annotations

import os

import numpy as np

import copy

from .data_io import load_pfm, load_rpc_as_array, load_rpc_as_qc_tensor

# offsets of the image-side normalisation inside the 170-vector (tools/RPCCore.py:8-28): LINE_OFF, SAMP_OFF, LINE_SCALE, SAMP_SCALE
_IMAGE_SIDE = (0, 1, 5, 6)


def read_img(filename):
    """PNG / any PIL-readable view -> PIL RGB image; single-band tiles are replicated to three bands (data_io.py:154-166)."""
    from PIL import Image
    org = Image.open(filename)
    bands = org.split()
    if len(bands) == 3:
        return org
    if len(bands) == 1:
        return Image.merge("RGB", (bands[0], bands[0], bands[0]))
    raise Exception("Images must have 3 channels or 1.")


def center_image(img):
    """(H,W,3) -> float32, per-channel zero mean and unit variance: (x - mean) / (sqrt(var) + 1e-8) (preprocess.py:150-160)."""
    x = np.array(img).astype(np.float32)
    var = np.var(x, axis=(0, 1), keepdims=True)
    mean = np.mean(x, axis=(0, 1), keepdims=True)
    return (x - mean) / (np.sqrt(var) + 0.00000001)


def scale_rpc(rpc, factor):
    """RPCs of an image decimated by `factor`: the image-side offsets and scales divided (satmvsdataset.py:80-90).  rpc (...,170)."""
    out = np.array(rpc, dtype=np.float64, copy=True)
    for i in _IMAGE_SIDE:
        out[..., i] = out[..., i] / factor
    return out


def decimate_nearest(a, factor):
    """cv2.resize(a, (w // factor, h // factor), interpolation=cv2.INTER_NEAREST) for an integer factor: destination pixel
    (y, x) takes source pixel (floor(y * h / (h // factor)), floor(x * w / (w // factor))) -- OpenCV's nearest rule."""
    h, w = a.shape[:2]
    oh, ow = h // factor, w // factor
    ys = np.minimum(np.floor(np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    return np.ascontiguousarray(a[ys][:, xs])


def _sample_paths(data_folder, view_num, ref_view, tile):
    """[ref.png, ref.rpc, src.png, src.rpc, ..., ref.pfm]: sources follow the ref view cyclically (gen_list.py:24-37)."""
    def path(kind, view, ext):
        return os.path.join(data_folder, "%s/%s/%s.%s" % (kind, view, tile, ext)).replace("\\", "/")
    sample = []
    for s in range(view_num):
        v = (ref_view + s) % view_num
        sample += [path("image", v, "png"), path("rpc", v, "rpc")]
    sample.append(path("height", ref_view, "pfm"))
    return sample


def gen_ref_list_rpc(data_folder, view_num, ref_view=2):
    """One sample per tile of image/<ref_view>/ (gen_list.py:45-84)."""
    folder = os.path.join(data_folder, "image/%s" % ref_view).replace("\\", "/")
    return [_sample_paths(data_folder, view_num, ref_view, os.path.splitext(p)[0]) for p in os.listdir(folder)]


def gen_all_mvs_list_rpc(data_folder, view_num):
    """Every view takes its turn as the reference (gen_list.py:6-42)."""
    out = []
    for r in range(view_num):
        out += gen_ref_list_rpc(data_folder, view_num, r)
    return out


def random_color(image, rng=np.random):
    """The reference's training-time augmentation (preprocess.py:163-178): colour, brightness, contrast and sharpness of a PIL image
    enhanced by factors drawn as randint / 100 from [0.01, 3], [0.1, 2], [0.1, 2], [0, 3], in that order."""
    from PIL import ImageEnhance
    image = ImageEnhance.Color(image).enhance(rng.randint(1, 301) / 100.)
    image = ImageEnhance.Brightness(image).enhance(rng.randint(10, 201) / 100.)
    image = ImageEnhance.Contrast(image).enhance(rng.randint(10, 201) / 100.)
    return ImageEnhance.Sharpness(image).enhance(rng.randint(0, 301) / 100.)


image_augment = random_color           # preprocess.py:156-160

_QC_IMAGE_SIDE = ("line_off", "samp_off", "line_scale", "samp_scale")


def scale_rpc_qc(rpcs, factor):
    """scale_rpc for a list of QC dictionaries (satmvsdataset.py:212-224)."""
    out = copy.deepcopy(rpcs)
    for r in out:
        for k in _QC_IMAGE_SIDE:
            r[k] = r[k] / factor
    return out


class MVSDataset:
    """Drop-in for dataset.satmvsdataset.MVSDataset (a torch Dataset: __len__ / __getitem__); modes "train", "val", "test",
    "pred"; `use_qc` selects the QC-dictionary samples (get_sample_qc / get_pred_sample_qc).  The "train" mode augments every view
    with the reference's random_color (drawing from np.random like the reference, or from RandomState(seed) when `seed` is given)
    unless another `augment` (a PIL image -> PIL image callable) is given; augment=False
    switches it off."""

    def __init__(self, data_folder, mode, view_num, ref_view=2, use_qc=False, augment=None, seed=None):
        assert mode in ["train", "val", "test", "pred"]
        self.data_folder, self.mode, self.view_num, self.ref_view, self.use_qc = data_folder, mode, view_num, ref_view, use_qc
        # seed: a reproducible stream for the default augmentation (np.random.RandomState(seed)) instead of the global np.random state the
        # reference draws from; with DataLoader workers pass a per-worker seed through worker_init_fn (every worker copies the dataset)
        self.rng = np.random.RandomState(seed) if seed is not None else None
        if augment is None and self.rng is not None:
            self.augment = lambda image: random_color(image, self.rng)
        else:
            self.augment = image_augment if augment is None else (augment or None)
        if mode == "pred" or ref_view < 0:
            self.sample_list = gen_all_mvs_list_rpc(data_folder, view_num)
        else:
            self.sample_list = gen_ref_list_rpc(data_folder, view_num, ref_view)
        self.sample_num = len(self.sample_list)

    def __len__(self):
        return len(self.sample_list)

    def _views(self, data, augment, qc=False):
        imgs, rpcs = [], []
        for view in range(self.view_num):
            image = read_img(data[2 * view])
            if augment is not None:
                image = augment(image)
            imgs.append(center_image(np.asarray(image)))
            rpcs.append(load_rpc_as_qc_tensor(data[2 * view + 1]) if qc else load_rpc_as_array(data[2 * view + 1])[0])
        if qc:
            cams = {"stage1": scale_rpc_qc(rpcs, 4), "stage2": scale_rpc_qc(rpcs, 2), "stage3": rpcs}
        else:
            rpcs = np.stack(rpcs)
            cams = {"stage1": scale_rpc(rpcs, 4), "stage2": scale_rpc(rpcs, 2), "stage3": rpcs}
        return np.stack(imgs).transpose([0, 3, 1, 2]), cams

    @staticmethod
    def _names(data):
        return data[0].split("/")[-2], os.path.splitext(data[0].split("/")[-1])[0]

    def get_sample(self, idx, qc=False):
        data = self.sample_list[idx]
        # (the reference indexes the height range by 2 * ref_view + 1 into the sample's path list, satmvsdataset.py:44 / :175 -- for
        #  the default ref_view = 2 of a 3-view sample that is the LAST source's rpc file, not the ref view's; reproduced)
        _, depth_max, depth_min = load_rpc_as_array(data[2 * self.ref_view + 1])
        depth_image = load_pfm(data[2 * self.view_num]).astype(np.float32)
        imgs, cams = self._views(data, self.augment if self.mode == "train" else None, qc)
        depth_values = np.array([depth_min, depth_max], dtype=np.float32)
        mask = np.float32((depth_image >= depth_min) * 1.0) * np.float32((depth_image <= depth_max) * 1.0)
        out_view, out_name = self._names(data)
        return {"imgs": imgs, "cam_para": cams,
                "depth": {"stage1": decimate_nearest(depth_image, 4), "stage2": decimate_nearest(depth_image, 2), "stage3": depth_image},
                "mask": {"stage1": decimate_nearest(mask, 4), "stage2": decimate_nearest(mask, 2), "stage3": mask},
                "depth_values": depth_values, "out_view": out_view, "out_name": out_name}

    def get_pred_sample(self, idx, qc=False):
        data = self.sample_list[idx]
        # (the pred samples take the height range of path 1 = the ref view's rpc in the 170-vector form, satmvsdataset.py:123, and
        #  of path 2 * ref_view + 1 in the QC form, :262; reproduced)
        _, depth_max, depth_min = load_rpc_as_array(data[2 * self.ref_view + 1 if qc else 1])
        imgs, cams = self._views(data, None, qc)
        out_view, out_name = self._names(data)
        return {"imgs": imgs, "cam_para": cams, "depth_values": np.array([depth_min, depth_max], dtype=np.float32),
                "out_view": out_view, "out_name": out_name}

    def get_sample_qc(self, idx):
        return self.get_sample(idx, qc=True)

    def get_pred_sample_qc(self, idx):
        return self.get_pred_sample(idx, qc=True)

    def __getitem__(self, idx):
        return self.get_pred_sample(idx, self.use_qc) if self.mode == "pred" else self.get_sample(idx, self.use_qc)
