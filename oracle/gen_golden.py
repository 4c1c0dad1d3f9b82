"""Generate tests/golden/*.npz -- runs ONLY in the build container, where /root/reference exists.

It imports the reference's own module 03b_irn/net/resnet50.py (the only hot-path module that
imports offline), drives it with seeded synthetic weights/inputs and stores
inputs + outputs as fixtures.  tests/test_oracle_golden.py then checks that oracle/cnn_ref.py
(the restatement that travels to the GPU box) reproduces them.  Nothing here is copied from the
reference: only numbers it computed, plus a slice of its cls_labels.npy data file.

    python oracle/gen_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/03b_irn"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import cnn_ref  # noqa: E402


def reference_resnet():
    sys.path.insert(0, REF)
    from net import resnet50 as ref_resnet50  # the reference's own file

    return ref_resnet50.ResNet(ref_resnet50.Bottleneck, [3, 4, 6, 3], strides=(2, 2, 2, 1))


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return h.hexdigest()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)

    sd = cnn_ref.make_resnet50_cam_state_dict(num_classes=20, seed=0)
    model = reference_resnet()
    missing, unexpected = model.load_state_dict(
        {k[len("resnet50."):]: v for k, v in sd.items() if k.startswith("resnet50.")}, strict=False)
    # num_batches_tracked buffers are the only keys the synthetic dict does not carry
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    assert not unexpected, unexpected
    model.eval()

    def ref_features(x):
        # the op sequence of resnet50_cam.CAM.forward's stage1..4 (resnet50_cam.py:57-63) on the
        # reference modules
        with torch.no_grad():
            y = model.maxpool(model.relu(model.bn1(model.conv1(x))))
            y = model.layer4(model.layer3(model.layer2(model.layer1(y))))
        return y

    fix = {"state_dict_sha256": np.array(sd_digest(sd))}
    rng = np.random.default_rng(20121)
    for S in (64, 97):
        img = cnn_ref.synth_image(rng, S + 11, S + 23)
        x = torch.from_numpy(cnn_ref.msf_pack(img, (S, S)))
        feat = ref_features(x)
        mine = cnn_ref.resnet50_features(x, sd)
        assert torch.equal(feat, mine), "restated ResNet50 differs from the reference module"
        with torch.no_grad():
            cam = torch.nn.functional.conv2d(feat, sd["classifier.weight"])
            cam = torch.relu(cam)
            cam = cam[0] + cam[1].flip(-1)
        fix["x_S%d" % S] = x.numpy()
        fix["feat_S%d" % S] = feat.numpy().astype(np.float32)
        fix["cam_S%d" % S] = cam.numpy()
        print("S=%d feat %s |feat| mean %.4f max %.3f cam max %.4f" %
              (S, tuple(feat.shape), feat.abs().mean(), feat.abs().max(), cam.max()))
    # a 321x321 case: checksums + slices only (the full tensors are large)
    img = cnn_ref.synth_image(rng, 375, 500)
    x = torch.from_numpy(cnn_ref.msf_pack(img, (321, 321)))
    feat = ref_features(x)
    assert torch.equal(feat, cnn_ref.resnet50_features(x, sd))
    cam = cnn_ref.resnet50_cam_forward(x, sd)
    fix["img_321"] = img
    fix["cam_321"] = cam.numpy()
    fix["feat_321_mean_abs"] = np.array(feat.abs().mean().item())
    fix["feat_321_slice"] = feat[:, ::64, ::5, ::5].numpy()
    print("S=321 feat %s mean|.| %.4f cam max %.4f" % (tuple(feat.shape), feat.abs().mean(), cam.max()))

    # make_cam tail on that CAM (make_cam.py:41-76), for two classes
    valid = torch.tensor([3, 11])
    s, h = cnn_ref.make_cam_tail(cam, (375, 500), valid)
    fix["tail_keys"] = valid.numpy()
    fix["tail_strided"] = s.numpy()
    fix["tail_highres_rows"] = h[:, ::25, :].numpy()
    fix["tail_highres_sum"] = np.array([h.double().sum().item(), (h.double() ** 2).sum().item()])

    # class labels of the first 256 train_aug images (reference data file voc12/cls_labels.npy,
    # image list voc12/train_aug.txt): pins valid_cat selection and the K distribution
    names = [l.strip() for l in open(os.path.join(REF, "voc12", "train_aug.txt"))][:256]
    cls = np.load(os.path.join(REF, "voc12", "cls_labels.npy"), allow_pickle=True).item()
    keys = [np.int32(int(n.split("_")[0]) * 1e6 + int(n.split("_")[1])) for n in names]
    fix["trainaug_names"] = np.array(names)
    fix["trainaug_labels"] = np.stack([cls[k] for k in keys]).astype(np.uint8)

    path = os.path.join(out_dir, "resnet50_cam.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
