#!/bin/bash
# Copies the outputs of tools/measure_r02.sh (merged back under gpurun_out/) into profiles/ under their committed names.  usage: collect_profiles.sh gpurun_out/r02p
o=${1:?measurement directory}
cp $o/bench.json profiles/r02_bench_default_output.json
cp $o/bench_prof.json profiles/r02_bench_no_cpu_baseline_output.json
cp $(ls $o/bench_prof/*/*kernel_stats.csv | head -1) profiles/r02_bench_default_kernel_stats.csv
grep '^{' $o/bench_share2.json > profiles/r02_bench_two_ranks_shared_gpu_rehearsal.json
for p in c_trainer_e2e:r02_c_trainer_end_to_end conv_bench:r02_conv_unet_shapes dp_step_floor:r02_dp_step_floor ew_bench:r02_elementwise_hbm_bandwidth gemm_sweep:r02_gemm_sweep_sizes_layouts \
         unet_blocks:r02_unet_blocks unet_model:r02_unet_model wsk_tiles:r02_wsk_tile_32_vs_16 unet_batch:r02_unet_batch; do
  grep -v amdgpu.ids $o/${p%%:*}.txt > profiles/${p##*:}.txt
done
cp $(ls $o/unet64_prof/*/*kernel_stats.csv | head -1) profiles/r02_unet_batch64_kernel_stats.csv
for t in conv128 conv256 conv8 convs2 mnist mnist_dp softmax_cols transpose add colsum rowsum; do
  cp $o/prof/$t.summary.json profiles/r02/$t.summary.json
  cp $(ls $o/prof/$t/stats/*/*kernel_stats.csv | head -1) profiles/r02/${t}_kernel_stats.csv
done
