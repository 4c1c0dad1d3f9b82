"""misc.indexing of the IRNet pipeline -- PathIndex and propagate_to_edge, the calls of
03b_irn/step/make_sem_seg_labels.py:59,76,93 (and cam_to_ir_label / train_irn's PathIndex).

The `misc` package is git-ignored in the reference tree; this follows upstream IRNet (jiwoon-ahn/irn,
misc/indexing.py).  The affinity step exists in-tree as AffinityDisplacementLoss.to_affinity
(net/vgg16_irn.py:247-261) and is what the path tables built here feed.  propagate_to_edge runs on the device
(csrc/rw.hip) as 2^exp_times applications of the sparse transition stencil instead of exp_times squarings of
the dense (hw x hw) matrix -- the same product, four orders of magnitude fewer operations."""
import numpy as np

from .. import _lib
from . import imutils


class PathIndex:
    """Search directions inside `radius` (upper half plane: dy > 0, or dy == 0 and dx > 0) and, for each, the
    pixels of the straight path from the source to the destination (both included), grouped by path length
    like upstream's `search_paths`; `search_dst` are the destinations in that grouped order."""

    def __init__(self, radius=5, default_size=None):
        self.radius = radius
        self.radius_floor = int(np.ceil(radius) - 1)
        self.search_paths, self.search_dst = self.get_search_paths_dst(self.radius)
        if default_size is not None:
            self.default_size = tuple(default_size)
            self.path_indices, self.src_indices, self.dst_indices = self.get_path_indices(default_size)

    def get_search_paths_dst(self, max_radius=5):
        coord_indices_by_length = [[] for _ in range(max_radius * 4)]
        search_dirs = []
        for x in range(1, max_radius):
            search_dirs.append((0, x))
        for y in range(1, max_radius):
            for x in range(-max_radius + 1, max_radius):
                if x * x + y * y < max_radius ** 2:
                    search_dirs.append((y, x))
        for dir in search_dirs:
            length_sq = dir[0] ** 2 + dir[1] ** 2
            path_coords = []
            min_y, max_y = sorted((0, dir[0]))
            min_x, max_x = sorted((0, dir[1]))
            for y in range(min_y, max_y + 1):
                for x in range(min_x, max_x + 1):
                    dist_sq = (dir[0] * x - dir[1] * y) ** 2 / length_sq
                    if dist_sq < 1:
                        path_coords.append([y, x])
            path_coords.sort(key=lambda c: -abs(c[0]) - abs(c[1]))
            coord_indices_by_length[len(path_coords)].append(path_coords)
        path_list_by_length = [np.asarray(v) for v in coord_indices_by_length if v]
        path_destinations = np.concatenate([p[:, 0] for p in path_list_by_length], axis=0)
        return path_list_by_length, path_destinations

    def get_path_indices(self, size):
        full_indices = np.reshape(np.arange(0, size[0] * size[1], dtype=np.int64), (size[0], size[1]))
        cropped_height = size[0] - self.radius_floor
        cropped_width = size[1] - 2 * self.radius_floor
        path_indices = []
        for paths in self.search_paths:
            path_indices_list = []
            for p in paths:
                coord_indices_list = []
                for dy, dx in p:
                    coord_indices = full_indices[dy:dy + cropped_height,
                                                 self.radius_floor + dx:self.radius_floor + dx + cropped_width]
                    coord_indices_list.append(np.reshape(coord_indices, [-1]))
                path_indices_list.append(coord_indices_list)
            path_indices.append(np.array(path_indices_list))
        src_indices = np.reshape(full_indices[:cropped_height, self.radius_floor:self.radius_floor + cropped_width], -1)
        dst_indices = np.concatenate([p[:, 0] for p in path_indices], axis=0)
        return path_indices, src_indices, dst_indices

    def device_tables(self):
        """(dirs int32 [D][2], path_start int32 [D+1], path_yx int32 [n][2]) for wsc_rw_propagate."""
        dirs, start, yx = [], [0], []
        for paths in self.search_paths:
            for p in paths:
                dirs.append(p[0])  # sorted farthest-first: the destination
                yx.extend(p.tolist())
                start.append(len(yx))
        return (np.asarray(dirs, np.int32).reshape(-1, 2), np.asarray(start, np.int32),
                np.asarray(yx, np.int32).reshape(-1, 2))


_PATH_CACHE = {}


def propagate_to_edge(x, edge, radius=5, beta=10, exp_times=8, ctx=None):
    """x (K,h,w) scores, edge (1,h,w) or (h,w) boundary map in [0,1] -> rw (K,1,h,w) as upstream returns.
    numpy or torch in, same kind out."""
    is_torch = hasattr(x, "detach")
    xn = np.ascontiguousarray(x.detach().cpu().numpy() if is_torch else x, dtype=np.float32)
    en = np.ascontiguousarray(edge.detach().cpu().numpy() if hasattr(edge, "detach") else edge, dtype=np.float32)
    K, h, w = xn.shape[-3], xn.shape[-2], xn.shape[-1]
    xn = xn.reshape(K, h, w)
    en = en.reshape(h, w)
    if radius not in _PATH_CACHE:
        _PATH_CACHE[radius] = PathIndex(radius=radius).device_tables()
    dirs, start, yx = _PATH_CACHE[radius]
    ctx = ctx or imutils.default_context()
    x_dev, e_dev = ctx.to_device(xn), ctx.to_device(en)
    rw_dev = _lib.rw_propagate(ctx, x_dev, e_dev, K, h, w, dirs, start, yx, float(beta), 2 ** int(exp_times))
    rw = ctx.to_host(rw_dev, (K, 1, h, w), np.float32)
    for b in (x_dev, e_dev, rw_dev):
        b.free()
    if is_torch:
        import torch

        return torch.from_numpy(rw)
    return rw


def device_path_tables(radius=5):
    """(dirs, path_start, path_yx) of PathIndex(radius) as wsc_rw_propagate_batch takes them (cached)."""
    if radius not in _PATH_CACHE:
        _PATH_CACHE[radius] = PathIndex(radius=radius).device_tables()
    return _PATH_CACHE[radius]


def propagate_to_edge_batch(xs, edges, radius=5, beta=10, exp_times=8, ctx=None):
    """propagate_to_edge for a list of images of different sizes in one device pass (every stencil step is one
    launch over all of them): xs[b] (K_b,h_b,w_b), edges[b] (1,h_b,w_b) or (h_b,w_b) -> list of (K_b,1,h_b,w_b)."""
    xs = [np.ascontiguousarray(x, dtype=np.float32) for x in xs]
    es = [np.ascontiguousarray(e, dtype=np.float32).reshape(x.shape[-2], x.shape[-1]) for x, e in zip(xs, edges)]
    if radius not in _PATH_CACHE:
        _PATH_CACHE[radius] = PathIndex(radius=radius).device_tables()
    dirs, start, yx = _PATH_CACHE[radius]
    ctx = ctx or imutils.default_context()
    Ks, hs, ws = [x.shape[0] for x in xs], [x.shape[1] for x in xs], [x.shape[2] for x in xs]
    x_dev = ctx.to_device(np.concatenate([x.ravel() for x in xs]))
    e_dev = ctx.to_device(np.concatenate([e.ravel() for e in es]))
    rw_dev = _lib.rw_propagate_batch(ctx, x_dev, e_dev, Ks, hs, ws, dirs, start, yx, float(beta), 2 ** int(exp_times))
    flat = ctx.to_host(rw_dev, (sum(k * h * w for k, h, w in zip(Ks, hs, ws)),), np.float32)
    for b in (x_dev, e_dev, rw_dev):
        b.free()
    out, off = [], 0
    for k, h, w in zip(Ks, hs, ws):
        out.append(flat[off:off + k * h * w].reshape(k, 1, h, w).copy())
        off += k * h * w
    return out

