"""DeepGlobe land-cover MSF classification dataset -- input contract of make_cam for the `deepglobe` /
`deepglobe_balanced` datasets (03b_irn/deepglobe/dataloader.py: CAT_LIST :14, label list :29-35 (the last
class 'unknown' is dropped), TorchvisionNormalize :60-84 (x / 255), DeepGlobeClassificationDatasetMSF
:188-226).  Items as in adp.dataloader (with "orig_img"); make_cam writes no "high_res" for this dataset."""
import os

import numpy as np

from ..adp.dataloader import find_cls_labels, load_img_name_list, msf_item

CAT_LIST = ["urban", "agriculture", "rangeland", "forest", "water", "barren", "unknown"]


def get_img_path(img_name, root):
    return os.path.join(root, "JPEGImages", img_name + ".jpg")


def load_image_label_list_from_npy(img_name_list, is_balanced, cls_labels_path=None):
    path = find_cls_labels(os.path.join("deepglobe", "cls_labels_balanced.npy" if is_balanced
                                        else "cls_labels_unbalanced.npy"), cls_labels_path)
    cls = np.load(path, allow_pickle=True).item()
    return np.array([cls[n][:-1] for n in img_name_list])


class TorchvisionNormalize:
    def __init__(self, norm_mode="int"):
        self.norm_mode = norm_mode
        if norm_mode == "int":
            self.mean, self.std = (0.0,) * 3, (255.0,) * 3
        elif norm_mode == "float":
            self.mean, self.std = (0.0,) * 3, (1.0,) * 3
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


class DeepGlobeClassificationDatasetMSF:
    def __init__(self, img_name_list_path, dev_root, is_balanced, norm_mode="float", outsize=None, scales=(1.0,),
                 cls_labels_path=None):
        assert norm_mode in ["float", "int"]
        assert outsize in [(321, 321), (224, 224), None]
        self.scales = tuple(scales)
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.dev_root = dev_root
        self.outsize = outsize
        self.norm = TorchvisionNormalize(norm_mode)
        self.label_list = load_image_label_list_from_npy(self.img_name_list, is_balanced, cls_labels_path)

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        from PIL import Image

        name = str(self.img_name_list[idx])
        img = np.asarray(Image.open(get_img_path(name, self.dev_root)).convert("RGB"))
        from ..misc import imutils

        items = [msf_item(si, self.outsize, self.norm) for si in imutils.scale_images(img, self.scales)]
        x = items[0][0] if len(items) == 1 else [it[0] for it in items]
        orig = items[0][1] if len(items) == 1 else [it[1] for it in items]
        return {"name": name, "img": x, "orig_img": orig, "size": (img.shape[0], img.shape[1]),
                "label": self.label_list[idx]}


class DeepGlobeImageDataset:
    """deepglobe/dataloader.py:117-176 as cam_to_ir_label uses it (`norm_mode=None, to_torch=False`): {"name", "img": uint8 HWC}."""

    def __init__(self, img_name_list_path, dev_root, is_balanced, norm_mode=None, to_torch=False, **augment):
        if any(v for v in augment.values()) or norm_mode is not None or to_torch:
            raise NotImplementedError("DeepGlobeImageDataset: only the plain image reader of the inference steps is provided")
        from ..adp.dataloader import load_img_name_list

        self.img_name_list = load_img_name_list(img_name_list_path)
        self.dev_root, self.is_balanced = dev_root, is_balanced

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        from PIL import Image

        name = self.img_name_list[idx]
        return {"name": name, "img": np.asarray(Image.open(get_img_path(name, self.dev_root)).convert("RGB"))}
