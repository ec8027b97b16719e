#!/bin/bash
# The profile set a round commits under profiles/ (run on the GPU box):   gpurun -- 'bash tools/profile_recipe.sh r2'
#   <tag>_train_kernel_stats.csv (+ .bench.json)     rocprofv3 --kernel-trace --stats of the training-step bench
#   <tag>_pmc_traffic.json                            HBM bytes per launch per kernel: separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   <tag>_pmc_sq_counters_summary.txt                 SQ counters per kernel (MFMA busy, waits, LDS) from their own --pmc pass
#   <tag>_predict_kernel_stats.csv, <tag>_pmc_traffic_predict.json   the same for one 512^3 12-direction prediction
# Counter passes run with --pmc only (no tracing domains), as the pool requires; the program sits directly behind `--`.
set -u
tag=${1:-r2}
out=gpurun_out/prof_$tag
mkdir -p $out profiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TRAIN="python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-predict --step-mode eager"
SHORT="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-predict --step-mode eager"
PRED="python3 tools/predict_once.py 128"
echo "[recipe] train kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -o t -- $TRAIN > $out/train_bench.json 2> $out/train.err
cp $out/train/t_kernel_stats.csv profiles/${tag}_train_kernel_stats.csv && cp $out/train_bench.json profiles/${tag}_train_kernel_stats.bench.json
echo "[recipe] train FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $SHORT > /dev/null 2> $out/fetch.err
echo "[recipe] train WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $SHORT > /dev/null 2> $out/write.err
python3 tools/pmc_traffic.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv profiles/${tag}_pmc_traffic.json > $out/traffic.txt 2>&1
echo "[recipe] train SQ counters"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -o s -- $SHORT > /dev/null 2> $out/sq.err
python3 tools/pmc_summary.py $out/sq/s_counter_collection.csv > profiles/${tag}_pmc_sq_counters_summary.txt 2> $out/sqsum.err
echo "[recipe] predict kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/pred -o p -- $PRED > $out/pred.txt 2> $out/pred.err
cp $out/pred/p_kernel_stats.csv profiles/${tag}_predict_kernel_stats.csv
echo "[recipe] predict FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pfetch -o f -- $PRED > /dev/null 2> $out/pfetch.err
echo "[recipe] predict WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pwrite -o w -- $PRED > /dev/null 2> $out/pwrite.err
python3 tools/pmc_traffic.py $out/pfetch/f_counter_collection.csv $out/pwrite/w_counter_collection.csv profiles/${tag}_pmc_traffic_predict.json > $out/ptraffic.txt 2>&1
echo "[recipe] per-layer traffic"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/lf -o f -- python3 tools/per_layer_traffic.py run > $out/lf.txt 2> $out/lf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/lw -o w -- python3 tools/per_layer_traffic.py run > $out/lw.txt 2> $out/lw.err
python3 tools/per_layer_traffic.py merge $out/lf/f_counter_collection.csv $out/lw/w_counter_collection.csv gpurun_out/layer_seq.json profiles/${tag}_per_layer_traffic.md > $out/layer_merge.txt 2>&1
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/
cp $out/*.txt $out/*.err gpurun_out/profiles_$tag/ 2>/dev/null
rm -rf $out      # the raw traces / counter CSVs are hundreds of MB; gpurun merges at most 64 MiB back
ls -la gpurun_out/profiles_$tag
