#!/bin/bash
# Round 5, visit d: paired tails of the two-row pass (probe builds): bitwise comparison of the outputs + time per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in nopair pair nopair_nv pair_nv; do tools/spec_rows_probe_$v /tmp/dump_$v.bin; done
for v in pair nopair_nv pair_nv; do cmp /tmp/dump_nopair.bin /tmp/dump_$v.bin && echo "outputs of $v == nopair (bit for bit)"; done
python3 - <<'PY'
import numpy as np
a=np.fromfile('/tmp/dump_nopair.bin',np.float32).reshape(-1,40); b=np.fromfile('/tmp/dump_pair.bin',np.float32).reshape(-1,40)
d=(a.view(np.uint32)!=b.view(np.uint32))
print('rows', len(a), 'rows differing', int(d.any(1).sum()), 'by column', d.sum(0).tolist())
if d.any():
    i=int(np.nonzero(d.any(1))[0][0]); print('first', i, a[i], b[i])
PY
for v in nopair pair nopair_nv pair_nv; do echo "==== $v"; tools/spec_rows_probe_$v | grep -A1 -E "^2 rows|wavefronts" | grep -v "^--"; done
