#!/bin/bash
# Round 5, visit ak: the whole GPU suite + smoke on the final library, then the evidence set r05_ak (tools/profile_round5.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/prior_exactness.jsonl
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/r05_ak_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
tools/profile_round5.sh r05_ak 2>&1 | tail -30
