#!/bin/bash
# Builds tools/spec_rows_probe (stage 1 of the speculative-evaluation study) against the built libsmz.so and prints the static
# instruction mix of every variant; run the binary on the GPU box:  tools/spec_rows_probe | tee gpurun_out/r05_spec_rows_probe.txt
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
cd $T && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 $EXTRA -save-temps=obj -o $T/spec_rows_probe $R/tools/spec_rows_probe.hip \
    -L$R/stochastic-muzero_amd -l:libsmz.so -Wl,-rpath,'$ORIGIN/../stochastic-muzero_amd' 2>/dev/null || exit 1
cp $T/spec_rows_probe $R/tools/spec_rows_probe
python3 - $T/spec_rows_probe-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
parts = re.split(r'\n(_Z5probeILi\d+EEv\w+):', s)
names = {"0": "2 rows, one branch", "1": "2 rows, two branches", "2": "2 x (2 rows, one branch)", "3": "4 rows, one branch", "4": "4 rows, per-row matrices"}
print("static instruction mix of the WHOLE probe kernel (weight staging + one loop body; loop body = one pass):")
for i in range(1, len(parts), 2):
    var = re.search(r'ILi(\d+)E', parts[i]).group(1)
    body = parts[i + 1].split('.Lfunc_end')[0]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//'))]
    valu = [l for l in lines if l.startswith('v_')]
    tail = parts[i + 1]
    g = lambda k: (re.search(r'; %s: (\d+)' % k, tail) or [None, '?'])[1]
    print("  %-28s VALU %4d (v_pk_fma %3d, dpp %3d, v_exp %2d, v_rcp/v_div %2d) | LDS %3d (ds_read_b128 %3d) | s_nop %3d | s_waitcnt %3d | VGPRs %s scratch %s" % (
        names[var], len(valu), sum(l.startswith('v_pk_fma') for l in valu), sum('row_' in l for l in valu),
        sum(l.startswith('v_exp') for l in valu), sum(l.startswith(('v_rcp', 'v_div_')) for l in valu),
        sum(l.startswith('ds_') for l in lines), sum(l.startswith('ds_read_b128') for l in lines),
        sum(l.startswith('s_nop') for l in lines), sum(l.startswith('s_waitcnt') for l in lines), g('NumVgprs'), g('ScratchSize')))
PY
rm -rf $T
