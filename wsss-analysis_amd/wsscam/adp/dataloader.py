"""ADP (Atlas of Digital Pathology) MSF classification dataset -- input contract of make_cam for the
`adp_morph` / `adp_func` datasets (03b_irn/adp/dataloader.py: CAT_LIST :14-18, get_img_path :39-44,
load_img_name_list :46-50, TorchvisionNormalize :64-88, ADPClassificationDatasetMSF :201-240).

Items are dicts {"name", "img": float32 (2,3,S,S) [orig, h-flip] normalised with the ADP constants,
"orig_img": uint8 (2,H0,W0,3) [orig, h-flip] (what common_cam.modify_by_htt needs), "size", "label"}.
PNG decode uses PIL (imageio is not in this image); the resize is voc12.dataloader.resize_bilinear_f64."""
import os

import numpy as np

from ..voc12.dataloader import resize_bilinear_f64

CAT_LIST = {
    "morph": ["E.M.S", "E.M.U", "E.M.O", "E.T.S", "E.T.U", "E.T.O", "E.P", "C.D.I", "C.D.R", "C.L", "H.E", "H.K",
              "H.Y", "S.M.C", "S.M.S", "S.E", "S.C.H", "S.R", "A.W", "A.B", "A.M", "M.M", "M.K", "N.P", "N.R.B",
              "N.R.A", "N.G.M", "N.G.W"],
    "func": ["G.O", "G.N", "T"],
}


def get_img_path(img_name, root, is_eval):
    return os.path.join(root, "PNGImagesSubset" if is_eval else "PNGImages", img_name + ".png")


def load_img_name_list(dataset_path):
    """np.loadtxt(dtype=str, comments='%') of the reference: one name per line, '%' starts a comment."""
    names = []
    for line in open(dataset_path):
        line = line.split("%", 1)[0].strip()
        if line:
            names.append(line)
    return np.array(names)


def find_cls_labels(rel_path, explicit=None):
    """Where the image-level label tables are looked for.  They are dataset annotation files the reference keeps
    next to its loaders ('voc12/cls_labels.npy', 'adp/cls_labels_<htt>.npy', 'deepglobe/cls_labels_*.npy', read
    relative to the working directory: e.g. adp/dataloader.py:33-37); this package does not ship copies.
    Order: the explicit path, $WSSCAM_CLS_LABELS_ROOT/<rel_path>, <cwd>/<rel_path>, this package's directory."""
    if explicit:
        return explicit
    cands = []
    if os.environ.get("WSSCAM_CLS_LABELS_ROOT"):
        cands.append(os.path.join(os.environ["WSSCAM_CLS_LABELS_ROOT"], rel_path))
    cands.append(rel_path)
    cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), rel_path))
    for c in cands:
        if os.path.exists(c):
            return c
    raise FileNotFoundError("label table %r not found (looked in %s); pass cls_labels_path=... or set "
                            "WSSCAM_CLS_LABELS_ROOT to the reference's 03b_irn directory" % (rel_path, cands))


def load_image_label_list_from_npy(img_name_list, htt_type, cls_labels_path=None):
    """adp/dataloader.py:33-37 reads 'adp/cls_labels_<htt>.npy' relative to the working directory."""
    path = find_cls_labels(os.path.join("adp", "cls_labels_" + htt_type + ".npy"), cls_labels_path)
    cls = np.load(path, allow_pickle=True).item()
    return np.array([cls[n] for n in img_name_list])


class TorchvisionNormalize:
    def __init__(self, norm_mode="int"):
        self.norm_mode = norm_mode
        if norm_mode == "int":
            self.mean, self.std = (193.09203,) * 3, (56.450138,) * 3
        elif norm_mode == "float":
            self.mean, self.std = (0.757,) * 3, (0.221,) * 3
        elif norm_mode is not None:
            raise ValueError("norm_mode value is not 'int' or 'float'")

    def __call__(self, img):
        img = np.float32(img)
        if self.norm_mode is None:
            return img
        proc = np.empty(img.shape, np.float32)
        for c in range(3):
            if self.norm_mode == "int":
                proc[..., c] = (img[..., c] - self.mean[c]) / self.std[c]
            else:
                proc[..., c] = (img[..., c] / 255.0 - self.mean[c]) / self.std[c]
        return proc


def msf_item(img_u8, outsize, norm):
    """(img (2,3,S,S) float32, orig_img (2,H0,W0,3) uint8) of one scale-1.0 image."""
    x = resize_bilinear_f64(img_u8, outsize) if outsize is not None else np.asarray(img_u8, np.float64)
    x = np.transpose(norm(x), (2, 0, 1))
    return (np.stack([x, np.flip(x, -1)], axis=0).astype(np.float32),
            np.stack([img_u8, np.flip(img_u8, -1)], axis=0))


class ADPClassificationDatasetMSF:
    def __init__(self, img_name_list_path, dev_root, htt_type, is_eval, norm_mode="float", outsize=None,
                 scales=(1.0,), cls_labels_path=None):
        assert norm_mode in ["float", "int"]
        assert outsize in [(321, 321), (224, 224), None]
        self.scales = tuple(scales)
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.dev_root = dev_root
        self.htt_type = htt_type
        self.is_eval = is_eval
        self.outsize = outsize
        self.norm = TorchvisionNormalize(norm_mode)
        self.label_list = load_image_label_list_from_npy(self.img_name_list, htt_type, cls_labels_path)

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        from PIL import Image

        name = str(self.img_name_list[idx])
        img = np.asarray(Image.open(get_img_path(name, self.dev_root, self.is_eval)).convert("RGB"))
        from ..misc import imutils

        items = [msf_item(si, self.outsize, self.norm) for si in imutils.scale_images(img, self.scales)]
        x = items[0][0] if len(items) == 1 else [it[0] for it in items]
        orig = items[0][1] if len(items) == 1 else [it[1] for it in items]
        return {"name": name, "img": x, "orig_img": orig, "size": (img.shape[0], img.shape[1]),
                "label": self.label_list[idx]}


class ADPImageDataset:
    """adp/dataloader.py:121-180 as cam_to_ir_label uses it (`norm_mode=None, to_torch=False`): {"name", "img": uint8 HWC}."""

    def __init__(self, img_name_list_path, dev_root, htt_type, is_eval, norm_mode=None, to_torch=False, **augment):
        if any(v for v in augment.values()) or norm_mode is not None or to_torch:
            raise NotImplementedError("ADPImageDataset: only the plain image reader of the inference steps is provided")
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.dev_root, self.htt_type, self.is_eval = dev_root, htt_type, is_eval

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        from PIL import Image

        name = self.img_name_list[idx]
        return {"name": name, "img": np.asarray(Image.open(get_img_path(name, self.dev_root, self.is_eval)).convert("RGB"))}
