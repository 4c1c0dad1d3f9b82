"""GPU: the N>1 path of make_cam with the REAL model object -- make_cam.run(n_gpus=2) spawns one worker process per
shard (torch.multiprocessing.spawn, make_cam.py:120-122); both workers are mapped onto GPU 0 here (a gpurun box has one
MI355X), each creates its own HIP context after the spawn, runs the overlapped host pipeline on its images[g::2] and
writes its own files.  The union must equal a single-process run bit for bit (same kernels, same batches per shard
do not matter: every image is processed independently)."""
import argparse
import os

import numpy as np
import pytest

from wsscam import _lib, synth
from wsscam.step import make_cam

pytestmark = pytest.mark.gpu


def _args(out_dir, packs, sd, n_gpus, ids=None):
    return argparse.Namespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                              use_cls=list(range(20)), model_id="resnet50", state_dict=sd, split="train_aug", dataset_obj=packs,
                              cam_out_dir=out_dir, outsize=(97, 97), n_gpus=n_gpus, cam_batch_images=3, cam_device_ids=ids,
                              cam_precision=_lib.PREC_F16X3, cam_weights_name="unused", norm_mode="int", val_list=None,
                              dev_root=None, cam_scales=(1.0,), class_names={"bg": ["background"], "fg": ["c%d" % i for i in range(20)]})


def test_make_cam_two_processes_one_gpu(tmp_path):
    rng = np.random.default_rng(3)
    sd = synth.resnet50_cam_state_dict(20, seed=1)
    packs = []
    for i in range(9):
        H0, W0 = [(40, 50), (50, 40), (33, 47)][i % 3]
        img = synth.synth_image(rng, H0, W0)
        lab = np.zeros(20, np.float32)
        lab[[i % 20, (3 * i + 1) % 20]] = 1
        if i == 4:
            lab[:] = 0  # an image with no positive class: the three-empty-arrays record (make_cam.py:86-88)
        packs.append({"name": "im%02d" % i, "img": synth.msf_pack(img, (97, 97), synth.TorchvisionNormalize("int")),
                      "size": (H0, W0), "label": lab})
    d1, d2 = str(tmp_path / "one"), str(tmp_path / "two")
    make_cam.run(_args(d1, packs, sd, 1))
    make_cam.run(_args(d2, packs, sd, 2, ids=[0, 0]))
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2)) == ["im%02d.npy" % i for i in range(9)]
    for f in sorted(os.listdir(d1)):
        a = np.load(os.path.join(d1, f), allow_pickle=True).item()
        b = np.load(os.path.join(d2, f), allow_pickle=True).item()
        assert list(a) == list(b) == ["keys", "cam", "high_res"]
        for k in a:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (f, k)
    e = np.load(os.path.join(d2, "im04.npy"), allow_pickle=True).item()
    assert all(v.shape == (0,) for v in e.values())
