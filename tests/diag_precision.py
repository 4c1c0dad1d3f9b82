"""Diagnostic (not a test): error statistics of the four precision modes on a 321x321 batch."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))
from oracle import cnn_ref
from wsscam import _lib
from wsscam.net import resnet50_cam
from wsscam.step import make_cam

sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
rng = np.random.default_rng(3)
sizes = [(375, 500), (500, 333), (281, 500), (500, 500)]
packs = []
for i, sz in enumerate(sizes):
    lb = np.zeros(20, np.float32); lb[rng.choice(20, 3, replace=False)] = 1
    packs.append({"name": "i%d" % i, "img": cnn_ref.msf_pack(cnn_ref.synth_image(rng, *sz), (321, 321)), "size": sz, "label": lb})
class Args: split = "train_aug"; dataset = "voc12"; cam_out_dir = None
refs = [cnn_ref.make_cam_image(torch.from_numpy(p["img"]), sd, p["size"], torch.from_numpy(p["label"])) for p in packs]
for name, prec in (("bf16", _lib.PREC_BF16), ("f16", _lib.PREC_F16), ("bf16x3", _lib.PREC_BF16X3), ("f16x3", _lib.PREC_F16X3)):
    m = resnet50_cam.CAM(None, "voc12", "", 20, None, precision=prec); m.load_state_dict(sd); m.cuda(0)
    outs = make_cam.process_batch(m, packs, Args, save=False)
    raw = m.forward_batch(np.stack([p["img"] for p in packs]))
    for o, r, p in zip(outs, refs, packs):
        d = np.abs(o["high_res"] - r["high_res"])
        agree = (o["high_res"].argmax(0) == r["high_res"].argmax(0)).mean()
        with torch.no_grad(): rc = cnn_ref.resnet50_cam_forward(torch.from_numpy(p["img"]), sd).numpy()
        print("%-7s max %.2e mean %.2e p99.9 %.2e argmax-agree %.5f" % (name, d.max(), d.mean(), np.quantile(d, 0.999), agree))
    rr = np.stack([cnn_ref.resnet50_cam_forward(torch.from_numpy(p["img"]), sd).detach().numpy() for p in packs])
    print("%-7s raw cam: max|d|/max %.2e  mean|d|/mean %.2e" % (name, np.abs(raw - rr).max() / rr.max(), np.abs(raw - rr).mean() / rr.mean()))
