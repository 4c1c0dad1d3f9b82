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


# ---- 8 workers on one host: the host side of CamPipeline with the device stubbed out --------------------------------------
class _StubBuf:
    """Page-locked buffer stand-in: numpy bytes with DeviceBuffer / HostBuffer's .view()."""

    def __init__(self, nbytes):
        self.a = np.zeros(int(nbytes), np.uint8)

    def view(self, shape, dtype, offset_bytes=0):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.a[offset_bytes:offset_bytes + n].view(dtype).reshape(shape)

    def free(self):
        pass


class _StubLane:
    class _Ctx:
        def sync(self):
            pass

        def close(self):
            pass

    def __init__(self, B, S, n_sc):
        import threading

        self.ctx = self._Ctx()
        self.pin_in = _StubBuf(B * n_sc * 2 * 3 * S * S * 4)
        self.x_view = self.pin_in.view((B, n_sc, 2, 3, S, S), np.float32)
        self.out_cap = 0
        self.pin_out = None
        self.free = threading.Event()
        self.free.set()

    def ensure_out(self, s_tot, h_tot):
        need = (max(s_tot, 1) + max(h_tot, 1)) * 4
        if need > self.out_cap:
            self.out_cap = int(need * 1.5) + (1 << 20)
            self.pin_out = _StubBuf(self.out_cap)

    def close(self):
        pass


def _stub_pipeline(world, out_dir, S):
    from wsscam.step.pipeline import CamPipeline

    class _Model:
        num_classes = 20

        def cam_size(self, s):
            return (s - 1) // 16 + 1

    class StubPipeline(CamPipeline):
        def _make_lane(self, device, batch_images, S_):
            return _StubLane(batch_images, S_, self.n_sc)

        def _copy_out(self, lane, s_tot, h_tot):
            pass

        def _device_step(self, lane, n, metas):  # the device's part of a batch: sizes only, outputs stay zero
            keys = [self.keys_fn(m, None) for m in metas]
            sizes = [tuple(int(v) for v in m["size"]) for m in metas]
            shapes, s_off, h_off, so, ho = [], [], [], 0, 0
            for k, (H, W) in zip(keys, sizes):
                h4, w4 = (H - 1) // 4 + 1, (W - 1) // 4 + 1
                shapes.append((len(k), h4, w4, H, W))
                s_off.append(so)
                h_off.append(ho)
                so += len(k) * h4 * w4
                ho += len(k) * H * W
            lane.ensure_out(so, ho)
            return keys, shapes, s_off, h_off, so

    class Args:
        dataset = "voc12"
        cam_out_dir = out_dir

    return StubPipeline(_Model(), 0, 4, S, keys_fn=lambda pack, score: np.nonzero(pack["label"])[0].astype(np.int64),
                        save_fn=lambda name, keys, sc, hc: make_cam._save(Args, name, keys, sc, hc), needs_score=False,
                        n_lanes=3, world=world)


class _HostWorkDataset:
    """Items cost real host work, like decode + float64 resize + normalise do (numpy releases the GIL for it)."""

    def __init__(self, n, S, seed):
        self.n, self.S, self.seed = n, S, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed + i)
        img = rng.random((96, 120, 3))
        for _ in range(6):
            img = np.sqrt(img * img + 1e-3)  # ~ the arithmetic of a resize pass, GIL released
        x = np.resize(img.astype(np.float32), (3, self.S, self.S))
        return {"name": "w%03d" % i, "img": np.stack([x, x[:, :, ::-1]]), "size": (90 + i % 7, 120),
                "label": np.eye(20, dtype=np.float32)[i % 20]}


class _StubDecodeDataset:
    """Decode STUBBED (one precomputed network input handed out for every item: no host arithmetic per image), outputs at
    the VOC size with K = 2 classes: what is left is the pipeline's own host work -- staging copies, views, the .npy writer."""

    def __init__(self, n, S):
        self.n = n
        x = np.zeros((3, S, S), np.float32)
        self.pair = np.stack([x, x])
        self.label = np.zeros(20, np.float32)
        self.label[[3, 11]] = 1

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"name": "w%03d" % i, "img": self.pair, "size": (375, 500), "label": self.label}


def _host_rank_main(rank, world, port, out_dir, n_items, S, result_file, stub_decode=False):
    import time

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shards = torchutils.split_dataset(_StubDecodeDataset(n_items, S) if stub_decode else _HostWorkDataset(n_items, S, 7), world)
        pipe = _stub_pipeline(world, out_dir, S)
        dist.barrier()
        t0 = time.perf_counter()
        pipe.run(shards[rank])
        dt = time.perf_counter() - t0
        pipe.close()
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's reduction: the slowest rank sets the time
        threads = torch.tensor([pipe.n_loaders + pipe.n_writers], dtype=torch.int64)
        dist.all_reduce(threads, op=dist.ReduceOp.SUM)
        if rank == 0:
            with open(result_file, "w") as fh:
                fh.write("%f %d %d %d\n" % (float(t.item()), int(threads.item()), pipe.n_loaders, pipe.n_writers))
    finally:
        dist.destroy_process_group()


def test_eight_rank_host_budget(tmp_path):
    """Eight workers of the make_cam host pipeline on ONE host (gloo, device stubbed): every worker sizes its loader / writer
    pools from its share of the cores (host_thread_budget: the eight pools together fit the machine instead of 8 x 16 threads),
    every image is written exactly once, and the aggregate host throughput of the eight workers is not below what one worker
    with the whole machine reaches (no collapse from oversubscription)."""
    from wsscam.step.pipeline import host_thread_budget

    cores = len(os.sched_getaffinity(0))
    b1, b8 = host_thread_budget(1, cores), host_thread_budget(8, cores)
    assert b1["n_loaders"] + b1["n_writers"] <= max(2, min(16, cores))
    assert 8 * (b8["n_loaders"] + b8["n_writers"]) <= max(16, 2 * cores)
    assert host_thread_budget(8, 128) == {"cores": 128, "cores_per_rank": 16, "n_loaders": 8, "n_writers": 8}
    assert host_thread_budget(8, 64) == {"cores": 64, "cores_per_rank": 8, "n_loaders": 5, "n_writers": 3}
    n_items, S = 96, 64
    times = {}
    for world in (1, 8):
        out = tmp_path / ("w%d" % world)
        out.mkdir()
        res = str(tmp_path / ("res%d.txt" % world))
        mp.spawn(_host_rank_main, nprocs=world, args=(world, _free_port(), str(out), n_items, S, res), join=True)
        assert sorted(os.listdir(out)) == ["w%03d.npy" % i for i in range(n_items)]
        t, nthreads, nl, nw = open(res).read().split()
        times[world] = float(t)
        assert int(nthreads) == world * (int(nl) + int(nw))
        assert (int(nl), int(nw)) == ((b1 if world == 1 else b8)["n_loaders"], (b1 if world == 1 else b8)["n_writers"])
    # same images, same cores: eight sharded workers must not collapse against one worker (process start-up of eight
    # interpreters on a shared 8-core container moves this ratio between 0.6 and 1.5 from run to run: the bound catches a
    # collapse, a tighter one would be a flaky test)
    assert n_items / times[8] >= 0.3 * n_items / times[1], times
    # What 8 GPUs at ~2000 images/s each ask of the host: 16 000 images/s through loaders -> lanes -> finishers
    # -> the REAL .npy writer at the VOC size (K = 2: 1.6 MB per file), decode stubbed.  A GPU node has >= 64 cores for its 8
    # GPUs, i.e. the requirement is 250 images/s per core.  The figure is PRINTED next to the requirement; the assertion only
    # catches a collapse (a twentieth of the requirement): on the 8-core build container the same run measured 175 ... 1500
    # images/s from one minute to the next (shared cores, page cache), so a tight bound here would be a flaky test, not
    # evidence.  No scaling curve exists on hardware yet -- this pins the host side's bookkeeping and order of magnitude only.
    n_big = int(os.environ.get("WSC_TEST_HOST_ITEMS", 32 * max(8, cores)))
    out = tmp_path / "voc8"
    out.mkdir()
    res = str(tmp_path / "res_voc8.txt")
    mp.spawn(_host_rank_main, nprocs=8, args=(8, _free_port(), str(out), n_big, S, res, True), join=True)
    assert len(os.listdir(out)) == n_big
    rec = np.load(str(out / "w000.npy"), allow_pickle=True).item()
    assert rec["high_res"].shape == (2, 375, 500) and rec["cam"].shape == (2, 94, 125)
    rate = n_big / float(open(res).read().split()[0])
    need = 8 * 2000.0 * cores / 64.0
    print("8-rank host pipeline, decode stubbed, real .npy writer at VOC size: %.0f images/s on %d cores "
          "(8 x 2000 images/s on a 64-core node = %.0f on this many cores)" % (rate, cores, need))
    assert rate >= 0.05 * need, (rate, need)
