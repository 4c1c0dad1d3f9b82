"""Per-layer table of one ResNet50-CAM forward from a rocprofv3 --kernel-trace database.

usage: python profiles/conv_layer_table.py <results.db> [samples=64] [size=321] [planes=1]
(planes = 2: the f16x3 trace -- 32-channel K-steps, two 16-bit planes per activation)
The launch order of wsc_net_forward_cam (csrc/net.hip run_backbone) is fixed, so the i-th conv kernel
after the nchw->nhwc4 layout kernel is the i-th entry of the list built here.
"""
import re
import sqlite3
import sys


def resnet50_layers(S):
    def o(h, k, s, p):
        return (h + 2 * p - k) // s + 1

    L = []
    h = o(S, 7, 2, 3)
    L.append(("stem 7x7 s2 3->64", h, 3, 64, 7))
    h = o(h, 3, 2, 1)
    inp = 64
    for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 1)), 1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            ho = o(h, 3, s, 1)
            n = "layer%d.%d." % (li, b)
            L.append((n + "conv1 1x1 %d->%d" % (inp, planes), h, inp, planes, 1))
            L.append((n + "conv2 3x3 %d s%d" % (planes, s), ho, planes, planes, 3))
            if b == 0:
                # conv3 and the 1x1 projection of the shortcut are one GEMM over the concatenated channels (csrc/net.hip)
                L.append((n + "conv3+proj 1x1 (%d+%d)->%d" % (planes, inp, planes * 4), ho, planes + inp, planes * 4, 1))
            else:
                L.append((n + "conv3 1x1 %d->%d +res" % (planes, planes * 4), ho, planes, planes * 4, 1))
            inp = planes * 4
            h = ho
    L.append(("CAM head 1x1 2048->20", h, 2048, 20, 1))
    return L


def n_dispatches(M, cin, cout, k, num_cus=256, planes=1):
    """Launches conv_igemm_launch makes for one generic layer (csrc/conv_igemm.hip tile choice): a layer on the
    256 x 256 tile whose grid is r * 256 + rem tiles with 0 < rem <= 128 is cut into the square kernel (r whole
    rounds) and a 128 x 128 remainder launch."""
    if cin % 64 or cout % 256:
        return 1
    nk = k * k * cin // (32 if planes == 2 else 64)
    blocks_sq = ((M + 255) // 256) * (cout // 256)
    if nk < (16 if planes == 2 else 4) or blocks_sq < 192:  # conv_igemm.hip tile choice (f16x3: square tile from 16 K-steps on)
        return 1
    rounds, rem = divmod(blocks_sq, num_cus)
    return 2 if rounds >= 1 and 0 < rem and rem * 2 <= num_cus else 1


def main(db, N=64, S=321, planes=1):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x / d.workgroup_size_x, d.group_segment_size "
                     "from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    start = [i for i, r in enumerate(rows) if "nchw_to_nhwc4" in r[0]][-1]
    convs = [r for r in rows[start:] if "conv_igemm_kernel" in r[0] or "stem_pool_kernel" in r[0] or "cam_head_kernel" in r[0]]
    layers = resnet50_layers(S)
    if "stem_pool_kernel" in convs[0][0]:  # f16x3: conv 7x7 + BN + ReLU + the 3x3/2 max-pool in one launch (csrc/stem_pool.hip)
        layers[0] = ("stem 7x7 s2 3->64 + maxpool (fused)",) + layers[0][1:]
    print("%-44s %8s %8s %9s %7s %6s" % ("layer", "us", "TFLOP/s", "act TB/s", "blocks", "LDS KB"))
    tot = 0.0
    pos = 0
    for (name, ho, cin, cout, k) in layers:
        M = N * ho * ho
        nd = n_dispatches(M, cin, cout, k, planes=planes)
        rs = convs[pos:pos + nd]
        pos += nd
        us = sum(r[2] - r[1] for r in rs) / 1000.0
        tot += us
        fl = 2.0 * M * cout * k * k * cin
        # 16-bit activations in + out (+ residual); weights ignored
        by = 2.0 * M * cout * (2 if "+res" in name else 1) + 2.0 * N * (ho * (2 if "s2" in name else 1)) ** 2 * cin
        if "fused" in name:  # NHWC4 input planes in, pooled planes out
            by = 8.0 * N * S * S + 2.0 * N * ((ho + 1) // 2) ** 2 * cout
        by *= planes
        print("%-44s %8.1f %8.0f %9.2f %7s %6s" % (name, us, fl / us / 1e6, by / us / 1e6, "+".join(str(r[3]) for r in rs),
                                                    "/".join("%.0f" % (r[4] / 1024.0) for r in rs)))
    print("conv kernels total %.1f us" % tot)


if __name__ == "__main__":
    main(sys.argv[1], *(int(v) for v in sys.argv[2:]))
