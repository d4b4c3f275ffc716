"""Image-file helpers of the data pipeline (reference data/util.py:15-96, 129-145).  The reference decodes with
cv2 (absent in this image); Pillow decodes the same 8-bit PNG/JPEG/BMP files, and the channel order is flipped to
cv2's BGR so everything downstream (BGR->RGB swap in the dataset, BGR PNG writers) sees what it saw before."""
import glob
import os
import random

import numpy as np

IMG_EXTENSIONS = (".jpg", ".JPG", ".jpeg", ".JPEG", ".png", ".PNG", ".ppm", ".PPM", ".bmp", ".BMP")


def is_image_file(filename):
    return filename.endswith(IMG_EXTENSIONS)


def _get_paths_from_images(path):
    assert os.path.isdir(path), f"{path} is not a valid directory"
    images = [os.path.join(d, f) for d, _, files in sorted(os.walk(path)) for f in sorted(files) if is_image_file(f)]
    assert images, f"{path} has no valid image file"
    return images


def get_image_paths(data_type, dataroot):
    """(sorted image paths, None) for an image folder; lmdb/memcached stores are not on the bin_stage4 path."""
    if dataroot is None:
        return None, None
    if data_type != "img":
        raise NotImplementedError(f"data_type [{data_type}] is not recognized.")
    return sorted(_get_paths_from_images(dataroot)), None


def glob_file_list(root):
    return sorted(glob.glob(os.path.join(root, "*")))


def imread_u8(path):
    """uint8 HWC BGR (or HW1 for grey) — what cv2.imread(path, IMREAD_UNCHANGED) yields for 8-bit files."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("L", "RGB", "RGBA"):
            im = im.convert("RGB")
        a = np.asarray(im)
    if a.ndim == 2:
        return a[:, :, None]
    if a.shape[2] == 4:                       # cv2 gives BGRA
        return np.ascontiguousarray(a[:, :, [2, 1, 0, 3]])
    return np.ascontiguousarray(a[:, :, ::-1])


def read_img(path, env=None, size=None, resize_scale=0):
    """float32 HWC BGR in [0,1], at most 3 channels (reference data/util.py:73-95).  resize_scale != 0 resamples
    bicubically to int(scale*W) x int(scale*H) first."""
    if env is not None:
        raise NotImplementedError("lmdb stores are not on the bin_stage4 path")
    img = imread_u8(path)
    if resize_scale != 0:
        from PIL import Image
        h, w = img.shape[:2]
        rgb = Image.fromarray(np.ascontiguousarray(img[:, :, ::-1] if img.shape[2] == 3 else img[:, :, 0]))
        rgb = np.asarray(rgb.resize((int(resize_scale * w), int(resize_scale * h)), Image.BICUBIC))
        img = rgb[:, :, ::-1] if rgb.ndim == 3 else rgb[:, :, None]
    img = img.astype(np.float32) / 255.
    return img[:, :, :3]


def read_img_seq(path):
    """Folder (or list of files) -> float tensor [T,3,H,W] RGB in [0,1] (reference data/util.py:98-112)."""
    import torch
    files = path if isinstance(path, (list, tuple)) else sorted(glob.glob(os.path.join(path, "*")))
    imgs = np.stack([read_img(f) for f in files], axis=0)[:, :, :, [2, 1, 0]]
    return torch.from_numpy(np.ascontiguousarray(np.transpose(imgs, (0, 3, 1, 2)))).float()


def augment(img_list, hflip=True, rot=True):
    """The same random (horizontal flip, vertical flip, transpose) applied to every HWC image of the list.  Three
    uniform draws in that order, each only when its switch is on (reference data/util.py:129-144)."""
    do_h = hflip and random.random() < 0.5
    do_v = rot and random.random() < 0.5
    do_t = rot and random.random() < 0.5
    out = []
    for img in img_list:
        img = img[::-1 if do_v else 1, ::-1 if do_h else 1, :]
        out.append(img.transpose(1, 0, 2) if do_t else img)
    return out
