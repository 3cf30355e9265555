#!/bin/bash
# round 3, call K: the whole GPU suite once + smoke() on the committed tree
set -o pipefail
out=gpurun_out/r3_k; mkdir -p $out
python -m pytest tests -m gpu -q > $out/tests_all.log 2>&1; echo "tests rc $?" | tee -a $out/tests_all.log
grep -E "passed|failed|FAILED|Error" $out/tests_all.log | tail -n 8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
