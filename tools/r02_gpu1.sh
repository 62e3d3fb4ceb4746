#!/bin/bash
# round-2 GPU call 1: parity suite (incl. the new full-size oracle comparisons), PMC calibration, baseline bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O/cal; cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal/fetch -- tools/bin/pmc_calibrate > $O/cal/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal/write -- tools/bin/pmc_calibrate > $O/cal/write.log 2>&1
python tools/pmc_calibration_table.py $O/cal > $O/calibration.txt 2>&1
cat $O/calibration.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-1500 $O/bench.json
