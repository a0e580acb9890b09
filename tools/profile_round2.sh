#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
# Round-2 evidence: bench lines, rocprofv3 kernel stats, PMC passes (own passes, kernel-trace only).  Run through gpurun.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --workload config2 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config2.json 2>> $O/bench.err
python bench.py --workload config4 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config4.json 2>> $O/bench.err
python bench.py --workload full1080p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 4 > $O/bench_full1080p.json 2>> $O/bench.err
python bench.py --workload full2160p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 4 > $O/bench_full2160p.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py $A > $O/stats_c3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c3_fetch -- python3 $R/bench.py $A > $O/pmc_c3_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c3_write -- python3 $R/bench.py $A > $O/pmc_c3_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_sq -- python3 $R/bench.py $A > $O/pmc_c3_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_c3_tcc -- python3 $R/bench.py $A > $O/pmc_c3_tcc.log 2>&1
B="--workload full2160p --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full2160 -- python3 $R/bench.py $B > $O/stats_full2160.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f2160_fetch -- python3 $R/bench.py $B > $O/pmc_f2160_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_f2160_write -- python3 $R/bench.py $B > $O/pmc_f2160_w.log 2>&1
cp $(ls $O/stats_c3/*/*kernel_stats.csv | head -1) $O/kernel_stats_config3.csv
cp $(ls $O/stats_full2160/*/*kernel_stats.csv | head -1) $O/kernel_stats_full2160p.csv
# keep the merge-back small: drop the per-dispatch traces of the PMC passes except the counter tables
find $O -name "*kernel_trace.csv" -path "*pmc_*" -delete
ls -la $O | head -40
