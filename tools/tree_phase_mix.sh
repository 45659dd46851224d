#!/bin/bash
# Static instruction mix of the tree-phase device functions (CPU only): tools/tree_phase_mix.sh
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -S --cuda-device-only -o $T/mix.s $R/tools/tree_phase_mix.hip 2>$T/err || { tail -20 $T/err; exit 1; }
python3 - $T/mix.s <<'PY'
import re, sys, collections
s = open(sys.argv[1]).read()
parts = re.split(r'\n(_Z\d+k_\w+):', s)
for i in range(1, len(parts), 2):
    body = parts[i + 1].split('.Lfunc_end')[0]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//'))]
    c = collections.Counter(l.split()[0] for l in lines)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    f64 = sum(v for k, v in c.items() if k.startswith('v_') and ('f64' in k))
    div = sum(v for k, v in c.items() if k.startswith(('v_div_', 'v_rcp', 'v_sqrt', 'v_rsq')))
    print("%-28s VALU %4d (f64 %3d, div/rcp/sqrt %3d) | LDS %3d | SALU %3d | branches %2d" % (
        re.sub(r'^_Z\d+', '', parts[i]).split('N3smz')[0][:28], valu, f64, div, sum(v for k, v in c.items() if k.startswith('ds_')),
        sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch'))),
        sum(v for k, v in c.items() if k.startswith(('s_cbranch', 's_branch')))))
    top = sorted(((k, v) for k, v in c.items() if k.startswith('v_')), key=lambda x: -x[1])[:12]
    print("      ", ", ".join("%s %d" % kv for kv in top))
PY
rm -rf $T
