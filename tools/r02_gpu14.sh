#!/bin/bash
# feature-search tests, the variants table and a kernel trace of the feature-search loop
bash $GRAFT_REPO_ROOT/tools/r02_gpu11.sh
sed -i 's#gpurun_out/r02l#gpurun_out/r02n#' $GRAFT_REPO_ROOT/tools/r02_gpu12.sh
bash $GRAFT_REPO_ROOT/tools/r02_gpu12.sh
