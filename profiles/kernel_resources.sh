#!/bin/bash
# usage: kernel_resources.sh <object.o> [name regex]  -- VGPR / AGPR / SGPR / LDS / scratch of every gfx950 kernel in a hipcc object
# (reads the code object's metadata note; runs without a GPU)
set -e
obj="$1"; re="${2:-.}"
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" "$tmp/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.co" | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
rx = re.compile(sys.argv[1])
cur = {}
rows = []
for line in txt.splitlines():
    line = line.strip()
    m = re.match(r'-?\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k in ('agpr_count','group_segment_fixed_size','private_segment_fixed_size','sgpr_count','vgpr_count','vgpr_spill_count','sgpr_spill_count','name'):
        if line.startswith('- .') and cur.get('name'): rows.append(cur); cur = {}
        cur[k] = v
    if k == 'wavefront_size' and cur.get('name'):
        rows.append(cur); cur = {}
if cur.get('name'): rows.append(cur)
seen=set()
for r in rows:
    n = r.get('name','')
    if n in seen or 'vgpr_count' not in r: continue
    seen.add(n)
    try: d = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()
    except Exception: d = n
    if not rx.search(d): continue
    print('%-4s v %-4s a %-4s s  lds %-7s scratch %-5s spill %s  %s' % (r.get('vgpr_count'), r.get('agpr_count','0'), r.get('sgpr_count'), r.get('group_segment_fixed_size'), r.get('private_segment_fixed_size'), r.get('vgpr_spill_count','0'), d[:150]))
" "$re"
rm -rf "$tmp"
