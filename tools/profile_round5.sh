#!/bin/bash
# Round-5 evidence: bench lines, rocprofv3 kernel stats, PMC passes (own passes, kernel-trace only).  Run through gpurun.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
( time python bench.py > $O/bench_config3.json 2> $O/bench_config3.err ) 2> $O/bench_config3.time; cat $O/bench_config3.time | tail -3
python bench.py --workload config2 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config2.json 2>> $O/bench.err
python bench.py --workload config4 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config4.json 2>> $O/bench.err
python bench.py --workload config4 --dataset-clips 1200 --host-clips > $O/bench_config4_dataset.json 2>> $O/bench.err
python bench.py --workload full2160p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 8 > $O/bench_full2160p.json 2>> $O/bench.err
python bench.py --workload full1080p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 8 > $O/bench_full1080p.json 2>> $O/bench.err
RELAX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 8 --workload config3 --dataset-clips 512 --clips-per-step 32 --stub-compute-ms 300 --loader-workers 8 \
    > $O/rehearsal_config3.json 2> $O/rehearsal_config3.err
RELAX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 8 --workload config4 --dataset-clips 2048 --clips-per-step 64 --stub-compute-ms 300 --loader-workers 8 \
    > $O/rehearsal_config4.json 2> $O/rehearsal_config4.err
cd /tmp && export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-other-workloads --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py $A > $O/stats_c3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c3_fetch -- python3 $R/bench.py $A > $O/pmc_c3_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c3_write -- python3 $R/bench.py $A > $O/pmc_c3_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_sq -- python3 $R/bench.py $A > $O/pmc_c3_sq.log 2>&1
B="--workload config2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py $B > $O/stats_c2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c2_sq -- python3 $R/bench.py $B > $O/pmc_c2_sq.log 2>&1
C="--workload full2160p --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full2160 -- python3 $R/bench.py $C > $O/stats_full2160.log 2>&1
cp $(ls -t $O/stats_c3/*/*kernel_stats.csv | head -1) $O/kernel_stats_config3.csv
cp $(ls -t $O/stats_c2/*/*kernel_stats.csv | head -1) $O/kernel_stats_config2.csv
cp $(ls -t $O/stats_full2160/*/*kernel_stats.csv | head -1) $O/kernel_stats_full2160p.csv
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -la $O | head -50
