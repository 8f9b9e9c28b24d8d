#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python tools/discrete_bench.py > $OUT/r03m_discrete_bench.json 2>/dev/null; cut -c1-200 $OUT/r03m_discrete_bench.json
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03n_suite.log 2>&1; grep -E "passed|failed" $OUT/r03n_suite.log | tail -3 | cut -c1-300
