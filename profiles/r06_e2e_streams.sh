#!/bin/bash
# end-to-end leg of bench.py, ring depth 2 / 3, with the host-time split of a step
out=gpurun_out/r06_e2e_ring.txt
: > $out
for r in 3 2; do
  echo "== WSC_BENCH_E2E_RING=$r" >> $out
  WSC_BENCH_E2E_TRACE=1 WSC_BENCH_E2E_RING=$r E2E_REPS=3 timeout 160 python profiles/e2e_probe.py >> $out 2>&1
done
cat $out
