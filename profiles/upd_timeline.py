"""Phase timeline of update_splat_kernel (A/B build, WSC_CRF_UPD_TIMELINE=<file>): the stamps thread 0 of every block takes with
s_memtime at the phase boundaries of the last splatting update of a call.  Prints the average length of every phase in
microseconds (s_memtime counts shader cycles on gfx950, MI355X_MICROARCH.md: ~2.1 GHz under this load), the average block lifetime and the number of blocks
alive at once.  usage: python profiles/upd_timeline.py <file> [<file> ...]"""
import sys

import numpy as np

NAMES = ["start -> DMA issued / landed", "-> descriptors in, first partials requested (FG)", "-> partial rows summed in LDS (FG)",
         "-> E in the Q stage (blur + slice, FG)", "-> trips done", "-> Gaussian slots written", "-> block end (bilateral slots written)"]
TICK_US = 1.0 / 2100.0  # shader clock under load (the kernel's wall time / the stamps' span confirms it to a few %)


def main():
    for path in sys.argv[1:]:
        t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
        t = t[t[:, 0] > 0]
        print("%s: %d blocks" % (path, len(t)))
        prev = t[:, 0].copy()
        for i in range(1, 8):
            ok = t[:, i] > 0
            if not ok.any():
                continue
            d = (t[ok, i] - prev[ok]) * TICK_US
            print("    %-58s %7.2f us avg  (p10 %6.2f  p90 %6.2f)" % (NAMES[i - 1], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
            prev[ok] = t[ok, i]
        life = (t[:, 7] - t[:, 0]) * TICK_US
        span = (t[:, 7].max() - t[:, 0].min()) * TICK_US
        print("    block lifetime %.2f us avg; kernel span %.1f us; blocks alive on average %.1f" % (life.mean(), span, life.sum() / span))


if __name__ == "__main__":
    main()
