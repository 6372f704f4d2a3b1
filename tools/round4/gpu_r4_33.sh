#!/bin/bash
# round 4, call 33: the attention tests with the 16- and 24-head cases (head permutation of the causal grid), the smoke entry
O=gpurun_out/r4_33; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fattn.py -m gpu -q -x 2>&1 | tail -3 | tee $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/summary.txt
