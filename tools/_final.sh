#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r3; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "^\.*\[parity\]|passed|failed|Error|assert" | sed "s/^\.*//" > $O/parity_all.txt
python __graft_entry__.py smoke 2>&1 | grep graft > $O/smoke.txt
python bench.py 2>&1 | grep "^{" > $O/bench_full.json
tail -2 $O/parity_all.txt; grep "free-running" $O/parity_all.txt | cut -c1-300; cat $O/smoke.txt; cut -c1-400 $O/bench_full.json
