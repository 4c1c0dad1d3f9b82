"""CPU, world_size 2, gloo: the N>1 path of the pipeline is image sharding with no data-path collective
(03b_irn/step/make_cam.py:120-122).  Two ranks run make_cam._work on their shards with the device batch
stubbed out; the union of their .npy files must be the whole dataset, disjoint, in the reference's
round-robin order, and bench.py's max-over-ranks timing reduction must agree on every rank."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wsscam.misc import torchutils
from wsscam.step import make_cam


class _FakeModel:
    """Stands in for the device network: make_cam._work only needs .cuda(), .ctx.sync()."""

    class _Ctx:
        def sync(self):
            pass

    def __init__(self):
        self.ctx = self._Ctx()
        self.device = None

    def cuda(self, device):
        self.device = device
        return self


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir, n_items):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dataset = [{"name": "img_%04d" % i, "img": np.zeros((2, 3, 8, 8), np.float32), "size": (9 + i % 3, 11),
                    "label": np.eye(20, dtype=np.float32)[i % 20]} for i in range(n_items)]
        shards = torchutils.split_dataset(dataset, world)

        class Args:
            split = "train_aug"
            dataset = "voc12"
            cam_out_dir = out_dir
            cam_batch_images = 4
            cam_pipeline = False  # the overlapped host pipeline needs a device; the shard bookkeeping is the same

        seen = []

        def fake_batch(model, packs, args, save=True):
            for p in packs:
                keys = np.nonzero(p["label"])[0].astype(np.int64)
                make_cam._save(args, p["name"], keys, np.full((1, 3, 3), rank, np.float32),
                               np.full((1,) + tuple(p["size"]), rank, np.float32))
                seen.append(p["name"])
            return []

        orig = make_cam.process_batch
        make_cam.process_batch = fake_batch
        try:
            model = _FakeModel()
            make_cam._work(rank, model, shards, Args)
        finally:
            make_cam.process_batch = orig
        assert model.device == rank
        # the only cross-rank traffic of the benchmark: barrier + MAX of the elapsed time
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == float(world)
        gathered = [None] * world
        dist.all_gather_object(gathered, seen)
        if rank == 0:
            assert gathered[0] == ["img_%04d" % i for i in range(0, n_items, world)]
            assert gathered[1] == ["img_%04d" % i for i in range(1, n_items, world)]
            flat = sorted(sum(gathered, []))
            assert flat == ["img_%04d" % i for i in range(n_items)]
    finally:
        dist.destroy_process_group()


def test_two_rank_image_sharding():
    n_items = 11
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_main, nprocs=2, args=(2, _free_port(), d, n_items), join=True)
        files = sorted(os.listdir(d))
        assert files == ["img_%04d.npy" % i for i in range(n_items)]
        for i, f in enumerate(files):
            rec = np.load(os.path.join(d, f), allow_pickle=True).item()
            assert set(rec) == {"keys", "cam", "high_res"}            # make_cam.py:80-82
            assert rec["keys"].dtype == np.int64 and list(rec["keys"]) == [i % 20]
            assert rec["high_res"].shape == (1, 9 + i % 3, 11)
            assert float(rec["cam"][0, 0, 0]) == float(i % 2)          # written by the rank that owns images[i::2]


def test_npy_layouts(tmp_path):
    """The three on-disk layouts of make_cam.py:80-88."""

    class Args:
        dataset = "voc12"
        cam_out_dir = str(tmp_path)

    make_cam._save(Args, "a", np.array([3, 7]), np.zeros((2, 4, 5), np.float32), np.zeros((2, 16, 20), np.float32))
    make_cam._save(Args, "b", np.zeros(0, np.int64), None, None)
    Args.dataset = "deepglobe"
    make_cam._save(Args, "c", np.array([1]), np.zeros((1, 4, 5), np.float32), np.zeros((1, 16, 20), np.float32))
    a = np.load(tmp_path / "a.npy", allow_pickle=True).item()
    b = np.load(tmp_path / "b.npy", allow_pickle=True).item()
    c = np.load(tmp_path / "c.npy", allow_pickle=True).item()
    assert list(a) == ["keys", "cam", "high_res"] and a["cam"].shape == (2, 4, 5)
    assert all(v.shape == (0,) for v in b.values()) and list(b) == ["keys", "cam", "high_res"]
    assert list(c) == ["keys", "cam"]
