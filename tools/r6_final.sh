#!/bin/bash
# round 6, final state: smoke(), PMC traffic passes + full evidence at the final sources
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_evidence.sh r6_final 2>&1 | tail -5
