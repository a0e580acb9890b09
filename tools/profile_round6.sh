#!/bin/bash
# Round-6 evidence: bench lines, rocprofv3 kernel stats, PMC passes (own passes, kernel-trace only), per-launch table of one ResNet-50 pass.
# Run through gpurun; summaries are copied into profiles/ by hand (tools/README.md).
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
( time python bench.py > $O/bench_config3.json 2> $O/bench_config3.err ) 2> $O/bench_config3.time; cat $O/bench_config3.time | tail -3
python bench.py --workload config2 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config2.json 2>> $O/bench.err
python bench.py --workload config4 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config4.json 2>> $O/bench.err
python bench.py --workload config4 --dataset-clips 1200 --host-clips > $O/bench_config4_dataset.json 2>> $O/bench.err
python bench.py --workload full2160p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 8 > $O/bench_full2160p.json 2>> $O/bench.err
python bench.py --workload full1080p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 8 > $O/bench_full1080p.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-other-workloads --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py $A > $O/stats_c3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c3_fetch -- python3 $R/bench.py $A > $O/pmc_c3_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c3_write -- python3 $R/bench.py $A > $O/pmc_c3_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_sq -- python3 $R/bench.py $A > $O/pmc_c3_sq.log 2>&1
B="--workload config2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py $B > $O/stats_c2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c2_sq -- python3 $R/bench.py $B > $O/pmc_c2_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c2_fetch -- python3 $R/bench.py $B > $O/pmc_c2_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c2_write -- python3 $R/bench.py $B > $O/pmc_c2_w.log 2>&1
C="--workload full2160p --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full2160 -- python3 $R/bench.py $C > $O/stats_full2160.log 2>&1
cp $(ls -t $O/stats_c3/*/*kernel_stats.csv | head -1) $O/kernel_stats_config3.csv
cp $(ls -t $O/stats_c2/*/*kernel_stats.csv | head -1) $O/kernel_stats_config2.csv
cp $(ls -t $O/stats_full2160/*/*kernel_stats.csv | head -1) $O/kernel_stats_full2160p.csv
cd $R
RELAX_OPTS=rn_fuse=1 bash tools/resnet_layers.sh r06_final 1024 > /dev/null 2>&1
cp gpurun_out/resnet_layers_r06_final.txt $O/resnet50_per_launch.txt
RELAX_OPTS=rn_fuse=0 bash tools/resnet_layers.sh r06_nofuse 1024 > /dev/null 2>&1
cp gpurun_out/resnet_layers_r06_nofuse.txt $O/resnet50_per_launch_unfused.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_stats.csv" -path "*/stats_*" -delete
ls -la $O | head -50
