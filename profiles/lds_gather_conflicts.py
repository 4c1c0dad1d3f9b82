"""Expected LDS bank-conflict ratio of the splat phase's row gathers (update_splat_kernel, tile_gather): every lane group of
LP = 6 lanes reads the 96-byte Q row of ANOTHER pixel of the tile (the slot's next entry), 10 rows per ds_read_b128.  The rows
are data (which pixels a lattice vertex collects), so their LDS slots are random; the script plays that against the lane
groups and the 16 x 16-byte slots of a bank row (MI355X_MICROARCH.md, LDS section) for several row pitches and for a
class-plane-major stage.  Result: 0.55-0.62 extra cycles per LDS cycle whatever the pitch -- the measured 0.27 of the whole
kernel (r06_pmc_crf.txt) is this instruction diluted by the kernel's conflict-free LDS traffic, not a layout that a re-pitch
would fix.

    python profiles/lds_gather_conflicts.py
"""
import random

# ds_read_b128: four lane groups of 16 lanes, one LDS cycle each when conflict-free
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[x + 32 for x in g] for g in GROUPS]


def conflict_ratio(pitch_q=6, LP=6, npix=256, trials=20000, plane_q=None, seed=0):
    rnd = random.Random(seed)
    extra = tot = 0
    for _ in range(trials):
        rows = [rnd.randrange(npix) for _ in range(64 // LP + 1)]
        for g in GROUPS:
            load = {}
            for lane in g:
                grp, l = divmod(lane, LP)
                if grp >= 64 // LP:
                    continue  # (the wave's last 64 % LP lanes idle)
                q = (l * plane_q + rows[grp]) if plane_q else (rows[grp] * pitch_q + l)  # 16-byte unit of the access
                load.setdefault(q % 16, set()).add(q)
            cyc = max(len(v) for v in load.values())
            extra += cyc - 1
            tot += cyc
    return extra / tot


if __name__ == "__main__":
    for p in (6, 7, 8, 9, 10):
        print("row-major stage, pitch %2d x 16 B: conflict cycles / LDS cycles = %.3f" % (p, conflict_ratio(p)))
    for pq in (257, 260, 261, 264):
        print("class-plane-major stage, plane stride %d x 16 B: %.3f" % (pq, conflict_ratio(plane_q=pq)))
