"""What the Keras-era drivers (02_cues/demo.py, 03c_hsn/demo.py) load from disk before their batch loops, for callers that
do not hand the loaded objects in: `settings.ini`, the classifier CNN of a session (`<sess_id>.h5` weights + `<sess_id>.mat`
thresholds under MODEL_ROOT/<dataset>_<model_type>/) and the image list + image-level labels of a split (the CSV files the
reference's `Dataset` class feeds to Keras' flow_from_dataframe, 02_cues/dataset.py:98-124).

The `.json` architecture files are not read: the device networks have the two fixed architectures of the reference's
torch mirrors (net/vgg16.py:44, net/m7.py:41), selected by `model_type` -- 'VGG16*' -> modified VGG16, 'M7*' / 'X1.7' -> M7.
Reading `.h5` needs h5py (net.common.keras_h5_weight_list); a missing file or module is an error, never a fallback."""
import configparser
import csv
import os

import numpy as np

VOC_CLASSES = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
DEEPGLOBE_CLASSES = ["urban", "agriculture", "rangeland", "forest", "water", "barren", "unknown"]
ADP_CLASSES = ["E.M.S", "E.M.U", "E.M.O", "E.T.S", "E.T.U", "E.T.O", "E.P", "C.D.I", "C.D.R", "C.L", "H.E", "H.K", "H.Y", "S.M.C",
               "S.M.S", "S.E", "S.C.H", "S.R", "A.W", "A.B", "A.M", "M.M", "M.K", "N.P", "N.R.B", "N.R.A", "N.G.M", "N.G.W", "G.O",
               "G.N", "T"]


def read_settings(path=None):
    """02_cues/demo.py:16-24 / 03c_hsn/demo.py: `../settings.ini` relative to the working directory (WSSCAM_SETTINGS overrides)."""
    path = path or os.environ.get("WSSCAM_SETTINGS", os.path.join("..", "settings.ini"))
    cfg = configparser.ConfigParser()
    if not cfg.read(path):
        raise FileNotFoundError("settings file %s not found (pass the loaded objects, or set WSSCAM_SETTINGS)" % path)
    data = cfg["Download Directory"]["data_dir"]
    if not os.path.isabs(data):  # relative to the working directory, as the reference's os.path.join(...) use of it is
        data = os.path.normpath(os.path.join(os.getcwd(), data))
    return {"DATA_ROOT": data, "MODEL_ROOT": os.path.join(data, cfg["Data Folders"]["model_cnn_dir"]),
            "CUES_ROOT": os.path.join(data, cfg["Data Folders"]["cues_dir"])}


def read_option(path, key, default):
    """An optional key of settings.ini's [Data Folders] section (keys the reference's file does not have: defaults apply)."""
    path = path or os.environ.get("WSSCAM_SETTINGS", os.path.join("..", "settings.ini"))
    cfg = configparser.ConfigParser()
    if not cfg.read(path):
        raise FileNotFoundError(path)
    return cfg["Data Folders"].get(key, default) if cfg.has_section("Data Folders") else default


class SetList:
    """The three attributes the drivers read off a Keras DataFrameIterator: `directory`, `filenames`, `data` (labels)."""

    def __init__(self, directory, filenames, data):
        self.directory, self.filenames, self.data = directory, list(filenames), np.asarray(data)

    def images(self, lo=0, hi=None):
        from PIL import Image

        return [np.asarray(Image.open(os.path.join(self.directory, f)).convert("RGB")) for f in self.filenames[lo:hi]]


class Dataset:
    """02_cues/dataset.py:5-124 / 03c_hsn/dataset.py:5-124 without the Keras generators: per split the image directory, the
    file names ('Patch Names' column) and the label matrix (the class-name columns) of
    <devkit>/ImageSets/Segmentation/<split>.csv, in file order (the evaluation generators are not shuffled; the drivers only
    read `.filenames` / `.data` of the training one).

    The two reference classes differ, selected by `layout`:
      * "cues" (02_cues/dataset.py): `database_dir` is ALWAYS <parent of the working directory>/database (:12 -- the class
        takes no directory argument; an explicit `database_dir` here is an override for callers with another tree);
        VOC2012 = VOCdevkit/VOC2012 with ['trainaug', 'val']; DeepGlobe names 'DeepGlobe_train75' / 'DeepGlobe_train37.5'
        (the drivers' 'DeepGlobe' / 'DeepGlobe_balanced' are accepted as aliases of the two);
      * "hsn" (03c_hsn/dataset.py): `database_dir` is the DATA_ROOT argument (:8, demo.py:88); VOC2012 =
        VOCdevkit/VOC_trainaug_val/VOC2012 with the 'val' split only (:57-59); DeepGlobe names 'DeepGlobe' (train75) /
        'DeepGlobe_balanced' (train37.5) (:84-89)."""

    def __init__(self, data_type="ADP", size=321, batch_size=16, database_dir=None, layout="cues"):
        if layout not in ("cues", "hsn"):
            raise ValueError("layout must be 'cues' or 'hsn'")
        if layout == "hsn" and database_dir is None:
            raise ValueError("the 03c_hsn Dataset takes its database_dir argument (DATA_ROOT), there is no default")
        self.data_type, self.size, self.batch_size, self.layout = data_type, size, batch_size, layout
        self.database_dir = database_dir or os.path.join(os.path.dirname(os.getcwd()), "database")
        if data_type == "ADP":
            self.devkit_dir = os.path.join(self.database_dir, "ADPdevkit", "ADPRelease1")
            self.sets, self.is_evals, self.class_names = ["valid", "test"], [True, True], list(ADP_CLASSES)
        elif data_type == "VOC2012":
            if layout == "hsn":
                self.devkit_dir = os.path.join(self.database_dir, "VOCdevkit", "VOC_trainaug_val", "VOC2012")
                self.sets, self.is_evals = ["val"], [True]
            else:
                self.devkit_dir = os.path.join(self.database_dir, "VOCdevkit", "VOC2012")
                self.sets, self.is_evals = ["trainaug", "val"], [False, True]
            self.class_names = list(VOC_CLASSES)
        elif "DeepGlobe" in data_type:
            self.devkit_dir = os.path.join(self.database_dir, "DGdevkit")
            names = {"DeepGlobe": "train75", "DeepGlobe_balanced": "train37.5"}
            if layout == "cues":
                names.update({"DeepGlobe_train75": "train75", "DeepGlobe_train37.5": "train37.5"})
            train = names.get(data_type)
            if train is None:
                raise ValueError("unknown DeepGlobe data_type %r" % data_type)
            self.sets, self.is_evals, self.class_names = [train, "test"], [False, True], list(DEEPGLOBE_CLASSES)
        else:
            raise ValueError("unknown data_type %r" % data_type)
        img_folder = "PNGImages" if data_type == "ADP" else "JPEGImages"
        self.set_gens = {}
        for s in self.sets:
            with open(os.path.join(self.devkit_dir, "ImageSets", "Segmentation", s + ".csv"), newline="") as f:
                rows = list(csv.DictReader(f))
            self.set_gens[s] = SetList(os.path.join(self.devkit_dir, img_folder), [r["Patch Names"] for r in rows],
                                       np.array([[float(r[c]) for c in self.class_names] for r in rows], dtype=np.float32)
                                       .reshape(len(rows), len(self.class_names)))


def load_model(model_dir, sess_id, model_type, dataset, device=0, precision=None):
    """02_cues/demo.py:104-124 / 03c_hsn/utilities.py build_model + load_thresholds + get_grad_cam_weights:
    (device CAM wrapper with the session's weights, alpha (F, C), final layer name, thresholds (1, C))."""
    import scipy.io

    from .cues import utilities as cu
    from .net import m7_cam, vgg16_cam
    from .net.common import keras_h5_weight_list, state_dict_from_keras_weights

    cls = vgg16_cam.CAM if "VGG16" in model_type else m7_cam.CAM
    thresholds = scipy.io.loadmat(os.path.join(model_dir, sess_id + ".mat")).get("optimalScoreThresh")
    weights = keras_h5_weight_list(os.path.join(model_dir, sess_id + ".h5"))
    num_classes = int(np.asarray(thresholds).shape[1])
    ds_tag = {"ADP": "adp_morph", "VOC2012": "voc12"}.get(dataset, "deepglobe")
    model = cls(None, ds_tag, model_type, num_classes, None, precision=precision)
    model.load_state_dict(state_dict_from_keras_weights(weights, model_type, cls.root, model.batchnorm, np.asarray(thresholds)[0]))
    model.cuda(device)
    img_size = 321 if model_type in ("VGG16", "VGG16bg") else 224
    final_layer = cu.find_final_layer(model)
    alpha = cu.get_grad_cam_weights(model, final_layer, np.zeros((1, img_size, img_size, 3)))
    return model, alpha, final_layer, np.asarray(thresholds)


def fgbg_sessions(model_dir, sess_id, fgbg_modes):
    """02_cues/demo.py:139-150: where the background model of a VOC2012 session lives."""
    out = {}
    for m in fgbg_modes:
        if m == "fg":
            out[m] = (model_dir, sess_id)
        elif "fg" in sess_id:
            out[m] = (model_dir.replace("fg", "bg"), sess_id.replace("fg", "bg"))
        else:
            out[m] = (model_dir.replace("fg", "") + "bg", sess_id.replace("fg", "") + "bg")
    return out
