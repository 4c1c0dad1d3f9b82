"""Per-layer matrix-pipe utilisation of one ResNet50-CAM forward from two rocprofv3 --pmc passes.

    python profiles/conv_pmc_table.py <set1_results.db> <set2_results.db>      (the f16x3 stack: two planes, 32-channel K-steps)
set 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
set 2: SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs): a 32x32x16 f16/bf16 MFMA holds its
SIMD's matrix pipe for 32 cycles (MI355X_MICROARCH.md), so this is the fraction of the 2.5 PFLOP/s dense peak.
"""
import sqlite3
import sys

from conv_layer_table import n_dispatches, resnet50_layers


def load(db):
    c = sqlite3.connect(db)
    d = {}
    for disp, name, dur, cn, cv in c.execute("select dispatch_id, name, duration, counter_name, counter_value from pmc_events"):
        e = d.setdefault(disp, {"name": name, "dur": dur})
        e[cn] = e.get(cn, 0) + cv
    return [d[k] for k in sorted(d)]


def main(db1, db2, N=64, S=321, planes=2):
    a, b = load(db1), load(db2)
    L = resnet50_layers(S)
    nds = [n_dispatches(N * ho * ho, cin, cout, k, planes=planes) for (_, ho, cin, cout, k) in L]
    a, b = a[-sum(nds):], b[-sum(nds):]
    if "stem_pool_kernel" in a[0]["name"]:  # conv 7x7 + BN + ReLU + max-pool in one launch (csrc/stem_pool.hip); per-wave columns: per 128-row-tile wave equivalent
        L[0] = ("stem 7x7 s2 + maxpool (fused)",) + L[0][1:]

    def merge(rows):  # a layer cut into two launches: add durations and counters
        out = {"dur": sum(r["dur"] for r in rows)}
        for r in rows:
            for kk, v in r.items():
                if kk not in ("name", "dur"):
                    out[kk] = out.get(kk, 0) + v
        return out

    print("%-36s %7s %9s %8s %9s %9s %8s %6s %6s" % ("layer", "us", "mfma_util", "mfma/wv", "valu/wave", "salu/wave",
                                                    "lds/wave", "wait%", "bank%"))
    busy = cyc_tot = 0.0
    pos = 0
    for (name, ho, cin, cout, k), nd in zip(L, nds):
        x, y = merge(a[pos:pos + nd]), merge(b[pos:pos + nd])
        pos += nd
        M = N * ho * ho
        cpad = (cout + 63) // 64 * 64
        bn = 128 if cpad % 128 == 0 else 64
        nw = ((M + 127) // 128) * (cpad // bn) * 4  # in units of 128-row-tile waves, whatever tile ran
        cyc = x["dur"] * 2.4 * 1024
        busy += x["SQ_VALU_MFMA_BUSY_CYCLES"]
        cyc_tot += cyc
        print("%-36s %7.1f %8.1f%% %8.0f %9.0f %9.0f %8.0f %6.1f %6.1f" % (
            name[:36], x["dur"] / 1e3, 100 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc, x["SQ_INSTS_MFMA"] / nw,
            x["SQ_INSTS_VALU"] / nw, y["SQ_INSTS_SALU"] / nw, x["SQ_INSTS_LDS"] / nw,
            100 * x["SQ_WAIT_INST_ANY"] / x["SQ_WAVE_CYCLES"],
            100 * y["SQ_LDS_BANK_CONFLICT"] / max(y["SQ_LDS_IDX_ACTIVE"], 1)))
    print("# whole stack: matrix pipe busy %.1f%% of SIMD-cycles" % (100 * busy / cyc_tot))


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main(sys.argv[1], sys.argv[2])
