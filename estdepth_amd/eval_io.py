"""On-disk formats either side of the streaming path: what eval_hybrid_seq.py reads and writes.

Reads   data/general_eval_seq.py:24-66,167-227: a scene directory in the ScanNet export layout
        (``rgb/<n>.jpg|png``, ``depth/<n>.png`` uint16 millimetres, ``pose/<n>.txt`` 4x4 camera-to-world) or the
        7-Scenes layout (``frame-%06d.color.png``, ``.depth.png``, ``.pose.txt``), natural-sorted, every
        ``frame_interval``-th frame, frames with non-finite poses dropped; the per-frame sample dictionary
        {'img' [1,3,H,W] float 0..255 RGB, 'img_raw' [1,H,W,3], 'dmap' [1,1,h,w] metres with invalid = 0,
        'dmask' [1,1,h,w] bool, 'cam_pose' [1,4,4], 'cam_intr' [1,3,3], 'img_path'}.
Writes  eval_hybrid_seq.py:194-257: ``<out>/<scene>/{init_depth,refined_depth,init_prob,refined_prob}/<name>.npy``
        as float16 (the colourised .jpg previews need OpenCV colour maps and are not produced).

The image has no OpenCV; decoding uses Pillow and the colour image is resized with a numpy restatement of
``cv2.resize(..., INTER_LINEAR)`` on half-pixel centres in float arithmetic.  OpenCV's 8-bit path uses 11-bit
fixed-point weights, so resized pixels may differ by 1 grey level: that step is "parity unpinned" (cv2 absent here).
Images already at the working size, the depth maps (never resized, general_eval_seq.py:191) and poses are exact.
Host code only.
"""
import glob
import os
import re

import numpy as np
import torch

SCANNET_FX = 577.87          # data/general_eval_seq.py:152-154, for 640x480
SCANNET_CX, SCANNET_CY = 319.5, 239.5


def natural_key(path):
    """natsort-style ordering of the file names used by the datasets (digits compare as integers)."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]


def scaled_intrinsics(image_size=(320, 256)):
    """general_eval_seq.py:151-163: the 640x480 ScanNet intrinsics rescaled to image_size = (width, height)."""
    k = torch.tensor([[SCANNET_FX, 0, SCANNET_CX], [0, SCANNET_FX, SCANNET_CY], [0, 0, 1]])
    k[0, :] *= image_size[0] / 640.
    k[1, :] *= image_size[1] / 480.
    return k.to(torch.float32)


def resize_bilinear_u8(img, width, height):
    """HxWx3 uint8 -> height x width x 3 uint8, half-pixel-centre bilinear with edge clamping."""
    h, w = img.shape[:2]
    if (w, h) == (width, height):
        return img
    def taps(n_out, n_in):
        c = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(c).astype(np.int64)
        f = c - i0
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f
    y0, y1, fy = taps(height, h)
    x0, x1, fx = taps(width, w)
    a = img.astype(np.float64)
    top = a[y0][:, x0] * (1 - fx)[None, :, None] + a[y0][:, x1] * fx[None, :, None]
    bot = a[y1][:, x0] * (1 - fx)[None, :, None] + a[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def prepare_seqs(scene_dir, interval, start_i=0, scannet_layout=True):
    """general_eval_seq.py:24-66: [{img_path, dmap_path, pose_path}] for every ``interval``-th frame with a finite pose."""
    if scannet_layout:
        imgs = sorted(glob.glob(os.path.join(scene_dir, "rgb", "*")), key=natural_key)
        dmaps = sorted(glob.glob(os.path.join(scene_dir, "depth", "*")), key=natural_key)
    else:
        imgs = sorted(glob.glob(os.path.join(scene_dir, "*.color.*")), key=natural_key)
        dmaps = sorted((p for p in glob.glob(os.path.join(scene_dir, "*.depth.*")) if "colored" not in p), key=natural_key)
    if not imgs or not dmaps:
        raise RuntimeError("no frames under %s" % scene_dir)
    img_ext, dmap_ext = os.path.splitext(imgs[0])[1], os.path.splitext(dmaps[0])[1]
    out = []
    for i in range(start_i, len(imgs), interval):
        idx = int(re.findall(r"\d+", os.path.basename(imgs[i]))[0])
        if scannet_layout:
            rec = {"img_path": "%s/rgb/%d%s" % (scene_dir, idx, img_ext),
                   "dmap_path": "%s/depth/%d%s" % (scene_dir, idx, dmap_ext),
                   "pose_path": "%s/pose/%d.txt" % (scene_dir, idx)}
        else:
            rec = {"img_path": "%s/frame-%06d.color%s" % (scene_dir, idx, img_ext),
                   "dmap_path": "%s/frame-%06d.depth%s" % (scene_dir, idx, dmap_ext),
                   "pose_path": "%s/frame-%06d.pose.txt" % (scene_dir, idx)}
        if np.all(np.isfinite(np.loadtxt(rec["pose_path"]))):
            out.append(rec)
    return out


def _read_image(path):
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB") if im.mode not in ("I;16", "I", "F", "L") else im)


class SequenceReader:
    """Indexable frame source with the reference's sample dictionary (general_eval_seq.py:167-227)."""

    def __init__(self, scene_dir, image_size=(320, 256), depth_min=0.01, depth_max=5.0, frame_interval=10,
                 start_i=0, scannet_layout=True):
        self.image_size, self.depth_min, self.depth_max = tuple(image_size), depth_min, depth_max
        self.cam_intr = scaled_intrinsics(self.image_size)
        self.seqs = prepare_seqs(scene_dir, frame_interval, start_i, scannet_layout)

    def __len__(self):
        return len(self.seqs)

    def __getitem__(self, index):
        rec = self.seqs[index]
        rgb = resize_bilinear_u8(np.array(_read_image(rec["img_path"])[..., :3]), *self.image_size)
        dmap = _read_image(rec["dmap_path"]).astype(np.float64) / 1000.            # millimetres -> metres
        pose = np.loadtxt(rec["pose_path"])
        dmask = (dmap >= self.depth_min) & (dmap <= self.depth_max) & np.isfinite(dmap)
        dmap[~dmask] = 0
        img_raw = torch.from_numpy(rgb).to(torch.float32)
        return {"img": img_raw.permute(2, 0, 1).unsqueeze(0), "img_raw": img_raw.unsqueeze(0),
                "dmap": torch.from_numpy(dmap).to(torch.float32)[None, None],
                "dmask": torch.from_numpy(dmask)[None, None],
                "cam_pose": torch.from_numpy(pose).to(torch.float32).unsqueeze(0),
                "cam_intr": self.cam_intr.unsqueeze(0), "img_path": rec["img_path"]}


OUTPUT_DIRS = {"init_depth": ("depth", 2), "refined_depth": ("depth", 0), "init_prob": "init_prob",
               "refined_prob": "fused_prob"}


def save_window_outputs(outputs, out_dir, rgb_basename, target=0, which=tuple(OUTPUT_DIRS)):
    """eval_hybrid_seq.py:194-257: float16 .npy dumps of one target frame; returns {kind: path}.
    Note the reference's naming: ``init_depth`` is the FUSED stereo depth ("depth", i, 2), ``init_prob`` the
    initial-head confidence, ``refined_*`` the 2D-refined depth and the fused confidence."""
    stem = os.path.splitext(os.path.basename(rgb_basename))[0]
    written = {}
    for kind in which:
        key = OUTPUT_DIRS[kind]
        key = (key[0], target, key[1]) if isinstance(key, tuple) else (key, target)
        d = os.path.join(out_dir, kind)
        os.makedirs(d, exist_ok=True)
        a = outputs[key]
        a = (a.squeeze(1) if kind.endswith("depth") else a.squeeze()).detach().cpu().numpy()
        path = os.path.join(d, stem + ".npy")
        np.save(path, np.float16(a))
        written[kind] = path
    return written


def write_synthetic_scene(scene_dir, imgs, dmaps_m, poses):
    """Test/demo helper: store frames in the ScanNet layout (png via Pillow; depth uint16 mm)."""
    from PIL import Image
    for sub in ("rgb", "depth", "pose"):
        os.makedirs(os.path.join(scene_dir, sub), exist_ok=True)
    for i, (im, dm, p) in enumerate(zip(imgs, dmaps_m, poses)):
        n = i * 10
        Image.fromarray(np.asarray(im, dtype=np.uint8)).save(os.path.join(scene_dir, "rgb", "%d.png" % n))
        Image.fromarray(np.asarray(np.rint(np.asarray(dm) * 1000.), dtype=np.uint16)).save(
            os.path.join(scene_dir, "depth", "%d.png" % n))
        np.savetxt(os.path.join(scene_dir, "pose", "%d.txt" % n), np.asarray(p, dtype=np.float64))
