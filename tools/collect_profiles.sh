#!/bin/bash
# usage (here, after the gpurun call of tools/prof_round.sh + tools/prof_dense.sh has been merged back): bash tools/collect_profiles.sh <rNN>
# copies the summaries the judge reads from gpurun_out/ (scratch) into profiles/ (tracked)
set -e
r=${1:?usage: collect_profiles.sh rNN}
cd "$(dirname "$0")/.."
cp gpurun_out/${r}_final/pmc_hera350_f32_stream.json profiles/pmc_hera350_f32_stream.json
cp gpurun_out/${r}_dense_f32/pmc_hera350_f32_shared.json profiles/
cp gpurun_out/${r}_dense_f64/pmc_hera350_f64_shared.json profiles/
cp gpurun_out/${r}_final/trace/bench_kernel_stats.csv profiles/${r}_final_hera350_f32_kernel_stats.csv
cp gpurun_out/${r}_dense_f32/trace/dense_kernel_stats.csv profiles/${r}_dense_hera350_f32_shared_kernel_stats.csv
cp gpurun_out/${r}_dense_f64/trace/dense_kernel_stats.csv profiles/${r}_dense_hera350_f64_shared_kernel_stats.csv
cp gpurun_out/${r}_final/pmc/fetch_counter_collection.csv profiles/${r}_final_pmc_fetch_counter_collection.csv
cp gpurun_out/${r}_final/pmc/write_counter_collection.csv profiles/${r}_final_pmc_write_counter_collection.csv
for c in sq tcc fetch write; do
  cp gpurun_out/${r}_dense_f32/pmc/${c}_counter_collection.csv profiles/${r}_dense_f32_pmc_${c}_counter_collection.csv
  cp gpurun_out/${r}_dense_f64/pmc/${c}_counter_collection.csv profiles/${r}_dense_f64_pmc_${c}_counter_collection.csv
done
if [ -d gpurun_out/${r}_multi ]; then
  cp gpurun_out/${r}_multi/trace/multi_kernel_stats.csv profiles/${r}_multi_rank_of_8_kernel_stats.csv
  for c in sq lds tcc ea fetch write tcp; do
    [ -f gpurun_out/${r}_multi/pmc/${c}_counter_collection.csv ] && cp gpurun_out/${r}_multi/pmc/${c}_counter_collection.csv profiles/${r}_multi_pmc_${c}_counter_collection.csv
  done
  [ -f gpurun_out/${r}_multi/summary.log ] && cp gpurun_out/${r}_multi/summary.log profiles/${r}_multi_rank_of_8_summary.log
fi
true
