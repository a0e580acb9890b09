#!/bin/bash
# round 5 evidence, part B: rocprofv3 kernel stats and PMC passes (own passes, kernel-trace only) of the bench commands (run through gpurun)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-other-workloads --no-measure-traffic"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py $A > $O/stats_c3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c3_fetch -- python3 $R/bench.py $A > $O/pmc_c3_f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c3_write -- python3 $R/bench.py $A > $O/pmc_c3_w.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_sq -- python3 $R/bench.py $A > $O/pmc_c3_sq.log 2>&1
B="--workload config2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-measure-traffic"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py $B > $O/stats_c2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c2_sq -- python3 $R/bench.py $B > $O/pmc_c2_sq.log 2>&1
cp $(ls -t $O/stats_c3/*/*kernel_stats.csv | head -1) $O/kernel_stats_config3.csv
cp $(ls -t $O/stats_c2/*/*kernel_stats.csv | head -1) $O/kernel_stats_config2.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete;
ls -la $O | head -40
