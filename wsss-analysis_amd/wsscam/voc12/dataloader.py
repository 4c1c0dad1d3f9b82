"""VOC12 MSF classification dataset -- input contract of make_cam
(03b_irn/voc12/dataloader.py: load_img_name_list :60-66, TorchvisionResize :68-78,
TorchvisionNormalize :80-106, VOC12ClassificationDatasetMSF :210-246).

Items are dicts {"name", "img": float32 (2,3,S,S) [orig, h-flip], "size": (H0,W0), "label"}.
JPEG decode uses PIL (imageio/cv2 are not in this image).  The float64 bilinear resize follows
cv2.resize's INTER_LINEAR half-pixel convention; cv2 itself is absent offline so bit parity of
this host-side resize is unpinned -- it is I/O, outside the device hot path."""
import os

import numpy as np

IMG_FOLDER_NAME = "JPEGImages"


def decode_int_filename(int_filename):
    s = str(int(int_filename))
    return s[:4] + "_" + s[4:]


def load_img_name_list(dataset_path):
    names = [l.strip() for l in open(dataset_path) if l.strip()]
    return np.array([np.int32(int(n.split("_")[0]) * 1e6 + int(n.split("_")[1])) for n in names])


def get_img_path(img_name, voc12_root):
    if not isinstance(img_name, str):
        img_name = decode_int_filename(img_name)
    return os.path.join(voc12_root, IMG_FOLDER_NAME, img_name + ".jpg")


def resize_bilinear_f64(img, out_hw):
    H, W = img.shape[:2]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    im = np.asarray(img, dtype=np.float64)
    if (H, W) == (oh, ow):
        return im
    ys = np.clip((np.arange(oh) + 0.5) * H / oh - 0.5, 0, H - 1)
    xs = np.clip((np.arange(ow) + 0.5) * W / ow - 0.5, 0, W - 1)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, H - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    wy = (ys - y0)[:, None, None]
    wx = (xs - x0)[None, :, None]
    top = im[y0][:, x0] * (1 - wx) + im[y0][:, x1] * wx
    bot = im[y1][:, x0] * (1 - wx) + im[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


def _cv2_linear_taps_u8(src, dst, clamp_taps):
    """Source index and the two 11-bit fixed-point coefficients of OpenCV's 8-bit INTER_LINEAR along one axis."""
    scale = 1.0 / (np.float64(dst) / np.float64(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if clamp_taps:  # columns: the window is moved inside and the fraction dropped
        lo, hi = s < 0, s >= src - 1
        s = np.where(lo, 0, np.where(hi, src - 1, s))
        f = np.where(lo | hi, np.float32(0), f)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)  # saturate_cast<short>: round half to even
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, c0, c1


def resize_bilinear_u8(img, out_hw):
    """cv2.resize(uint8 image, (w, h)) with the default INTER_LINEAR, as read_batch of 02_cues/utilities.py:172-176 and
    03c_hsn/utilities.py:176-181 get it (the batch array is uint8).  OpenCV's 8-bit path is fixed point -- 11-bit
    coefficients, an int horizontal pass, the vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 --
    and an exact 2 x 2 decimation is the rounded box mean (INTER_AREA shortcut); restated from the published algorithm
    (cv2 is absent offline: parity against cv2 itself unpinned).  Bit-identical to wsc_resize_u8 (csrc/input.hip)."""
    im = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = im.shape[:2]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if (H, W) == (oh, ow):
        return im.copy()
    v = im.astype(np.int64)
    if (H, W) == (2 * oh, 2 * ow):
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _cv2_linear_taps_u8(W, ow, True)
    sy, b0, b1 = _cv2_linear_taps_u8(H, oh, False)
    x1 = np.minimum(sx + 1, W - 1)
    y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    hor = v[:, sx] * a0[None, :, None] + v[:, x1] * a1[None, :, None]  # (H, ow, 3) int
    out = (((b0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class TorchvisionNormalize:
    def __init__(self, norm_mode="int"):
        self.norm_mode = norm_mode
        if norm_mode == "int":
            self.mean, self.std = (104.0, 117.0, 123.0), (255.0, 255.0, 255.0)
        elif norm_mode == "float":
            self.mean, self.std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
        elif norm_mode is not None:
            raise ValueError("norm_mode value is not 'int' or 'float'")

    def __call__(self, img):
        proc = np.empty(img.shape, np.float32)
        img = np.float32(img)
        for c in range(3):
            if self.norm_mode == "int":
                proc[..., c] = (img[..., c] - self.mean[c]) / self.std[c]
            elif self.norm_mode == "float":
                proc[..., c] = (img[..., c] / 255.0 - self.mean[c]) / self.std[c]
            else:
                return img
        return proc


def msf_pack(img_u8, outsize, norm):
    x = norm(resize_bilinear_f64(img_u8, outsize) if outsize is not None else np.asarray(img_u8, np.float64))
    x = np.transpose(x, (2, 0, 1))
    return np.stack([x, np.flip(x, -1)], axis=0).astype(np.float32)


class VOC12ClassificationDatasetMSF:
    def __init__(self, img_name_list_path, dev_root, norm_mode="float", outsize=None, scales=(1.0,),
                 cls_labels_path=None, device_transform=False):
        assert norm_mode in ["float", "int"]
        assert outsize in [(321, 321), (224, 224), None]
        self.scales = tuple(scales)
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.dev_root = dev_root
        self.outsize = outsize
        self.norm = TorchvisionNormalize(norm_mode)
        # device_transform: items carry the decoded image ("img_u8") instead of the float32 pair ("img"); make_cam's
        # pipeline then runs resize + normalise + flip on the GPU (wsc_msf_input_u8, bit-identical to msf_pack)
        self.device_transform = device_transform and outsize is not None
        from ..adp.dataloader import find_cls_labels

        cls = np.load(find_cls_labels(os.path.join("voc12", "cls_labels.npy"), cls_labels_path),
                      allow_pickle=True).item()
        self.label_list = np.array([cls[n] for n in self.img_name_list])

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        from PIL import Image

        name_str = decode_int_filename(self.img_name_list[idx])
        img = np.asarray(Image.open(get_img_path(name_str, self.dev_root)).convert("RGB"))
        # voc12/dataloader.py:231-242: every scale is rescaled (PIL bicubic), then resized to `outsize` like the plain image;
        # one scale -> the array itself, several -> a list (the reference's item format)
        from ..misc import imutils

        s_imgs = imutils.scale_images(img, self.scales)
        if self.device_transform:
            return {"name": name_str, "img_u8": s_imgs[0] if len(s_imgs) == 1 else s_imgs, "size": (img.shape[0], img.shape[1]),
                    "label": self.label_list[idx]}
        ms = [msf_pack(si, self.outsize, self.norm) for si in s_imgs]
        return {"name": name_str, "img": ms[0] if len(ms) == 1 else ms, "size": (img.shape[0], img.shape[1]),
                "label": self.label_list[idx]}


class VOC12ImageDataset:
    """voc12/dataloader.py:137-190 as cam_to_ir_label uses it (`norm_mode=None, to_torch=False`, cam_to_ir_label.py:100-101):
    items {"name", "img": uint8 HWC}.  The training-time augmentations of the class (resize_long, rescale, crops, flips)
    belong to train_cam / train_irn, which are out of scope."""

    def __init__(self, img_name_list_path, dev_root, norm_mode=None, to_torch=False, **augment):
        if any(v for v in augment.values()) or norm_mode is not None or to_torch:
            raise NotImplementedError("VOC12ImageDataset: only the plain image reader of the inference steps is provided")
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.dev_root = dev_root

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        from PIL import Image

        name_str = decode_int_filename(self.img_name_list[idx])
        return {"name": name_str, "img": np.asarray(Image.open(get_img_path(name_str, self.dev_root)).convert("RGB"))}
