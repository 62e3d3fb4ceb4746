#!/bin/bash
# PMC passes over the warm-started iteration kernel
bash $GRAFT_REPO_ROOT/tools/pmc.sh r02pmc python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 --warm 2 --steps 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/pmc_summary.py gpurun_out/r02pmc k_warm > gpurun_out/r02pmc/summary.txt 2>&1; cat gpurun_out/r02pmc/summary.txt | head -70
find gpurun_out/r02pmc -name "*.csv" -size +2M -delete
