"""HBM traffic per launch of a kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    WSC_PROFILE_ROUND=r04 python profiles/make_traffic_json.py <fetch_results.db> <write_results.db> > profiles/hbm_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB
summed over the XCD instances of a dispatch, and FETCH_SIZE under-reports by 2x on gfx950 (doubled here).
bench.py copies `classes[<roofline kernel>].bytes_per_launch` into `roofline.traffic`.
"""
import json
import sqlite3
import sys
from collections import defaultdict

CLASSES = {  # bench.py kernel-class label -> (kernel-name substrings, substring whose dispatches count as launches)
    "update_splat_kernel": (("update_splat_kernel",), "update_splat_kernel"),
    "combine4+blur_lds+blur4+blur3_tile": (("combine4_kernel", "combine4_balanced_kernel", "blur_lds_kernel", "blur4_kernel", "blur3_tile_kernel"),
                                  ("combine4_kernel", "combine4_balanced_kernel", "blur_lds_kernel", "blur4_kernel", "blur3_tile_kernel")),
    "gauss_msg_kernel": (("gauss_msg_kernel",), "gauss_msg_kernel"),
    "blur3_tile_kernel": (("blur3_tile_kernel",), "blur3_tile_kernel"),
    # the splatting update with messages: round 6's message-in-update form (template argument FG = 4 / 6) and the two-launch form (0)
    "update_splat_kernel<true, true, true, true, 4>": (("update_splat_kernel<true, true, true, true, 4>",), "update_splat_kernel<true, true, true, true, 4>"),
    "update_splat_kernel<true, true, true, true, 0>": (("update_splat_kernel<true, true, true, true, 0>",), "update_splat_kernel<true, true, true, true, 0>"),
    # the conv stack: every instantiation together (bench.py's top-level roofline kernel), then per tile
    # shape -- the labels are bench.py's profile classes -- and the fused stem + max-pool kernel
    "conv_igemm_kernel": (("conv_igemm_kernel", "cam_head_kernel"), ("conv_igemm_kernel", "cam_head_kernel")),
    "conv_igemm_kernel<256-row tiles,glds>": (("conv_igemm_kernel<256, ",), "conv_igemm_kernel<256, "),
    "conv_igemm_kernel<128x128,glds>": (("conv_igemm_kernel<128, 128, ",), "conv_igemm_kernel<128, 128, "),
    "conv_igemm_kernel<128x64,glds>": (("conv_igemm_kernel<128, 64, ", "cam_head_kernel"), ("conv_igemm_kernel<128, 64, ", "cam_head_kernel")),
    "conv_igemm_kernel<small-Cin> / stem_pool_kernel": (("stem_pool_kernel",), "stem_pool_kernel"),
    "crf_build(all)": (("tile_embed", "slice_norm_tile", "tile_slots", "neighbors_kernel", "slot_dest", "scan_", "slot_ones", "assign_rows",
                        "first_bits", "combine1_kernel", "blur1_kernel", "fill_tables", "pack_pixels", "tile_build"), "tile_embed_kernel<5>"),
}


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    tot = defaultdict(float)
    disp = defaultdict(set)
    for name, d, cn, cv in c.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        if cn == counter:
            tot[name] += cv
            disp[name].add(d)
    return tot, {k: len(v) for k, v in disp.items()}


import os

ROUND = os.environ.get("WSC_PROFILE_ROUND", "r04")


def main(fetch_db, write_db):
    f, fd = per_kernel(fetch_db, "FETCH_SIZE")
    w, wd = per_kernel(write_db, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0 "
                     "--no-cpu-baseline --no-pipeline --quick; FETCH_SIZE doubled (gfx950 correction)", "round": ROUND, "classes": {}}
    for label, (subs, launch_sub) in CLASSES.items():
        fb = sum(v for k, v in f.items() if any(s in k for s in subs)) * 1024.0 * 2.0
        wb = sum(v for k, v in w.items() if any(s in k for s in subs)) * 1024.0
        ls = (launch_sub,) if isinstance(launch_sub, str) else launch_sub
        nf = sum(v for k, v in fd.items() if any(x in k for x in ls))
        nw = sum(v for k, v in wd.items() if any(x in k for x in ls))
        if nf == 0 or nw == 0:
            continue
        out["classes"][label] = {"read_bytes_per_launch": round(fb / nf), "write_bytes_per_launch": round(wb / nw),
                                 "bytes_per_launch": round(fb / nf + wb / nw), "launches_profiled": nf, "round": ROUND}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
