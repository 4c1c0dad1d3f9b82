#!/bin/bash
# Same-box A/B of two builds of the library: put the two .so files under ab_tmp/ (untracked, travels with gpurun) and run
#   gpurun -- 'bash profiles/lib_ab.sh ab_tmp/lib_base.so ab_tmp/lib_new.so'
# Alternates base / new twice: pipelined ms per step, then (no pipeline) step, lattice build, mean-field loop, update_splat us, build us.
cp wsss-analysis_amd/wsscam/libwsscam.so /tmp/lib_ship.so
trap 'cp /tmp/lib_ship.so wsss-analysis_amd/wsscam/libwsscam.so' EXIT  # the shipped library comes back whatever happens
for rep in 1 2; do for v in "$1" "$2"; do
  cp "$v" wsss-analysis_amd/wsscam/libwsscam.so
  echo "== $v"
  python bench.py --no-cpu-baseline --quick 2>&1 | tail -1 | cut -c88-170
  python bench.py --no-cpu-baseline --quick --no-pipeline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['stages']['kernels']; print(d['ms_per_step'], d['stages']['crf_create_ms'], d['stages']['crf_infer_ms'], k['update_splat_kernel']['avg_us'], k['crf_build(all)']['avg_us'])"
done; done
