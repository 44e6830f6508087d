#!/bin/bash
# PMC traffic of the factored path at the cfg-3 shape (separate FETCH_SIZE / WRITE_SIZE passes; the factored part of tools/regen_profiles.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcf; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PROBE_FACTORED=1 PROBE_B=64 PROBE_N=1000 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_f -o fetch -- python $R/tools/pmc_probe.py > $O/pmc_fetch_f.log 2>&1
PROBE_FACTORED=1 PROBE_B=64 PROBE_N=1000 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_f -o write -- python $R/tools/pmc_probe.py > $O/pmc_write_f.log 2>&1
cd $R
FF=$(find $O/pmc_fetch_f -name "*results.db" | head -1); WF=$(find $O/pmc_write_f -name "*results.db" | head -1)
python tools/pmc_summary.py $FF $WF $O/pmc_traffic_factored.json 64,1000,3 > $O/pmc_hbm_traffic_factored.txt 2>&1
cat $O/pmc_hbm_traffic_factored.txt
rm -rf $O/pmc_fetch_f $O/pmc_write_f
