#!/bin/bash
# Round-3 measurement pass on one MI355X (run through gpurun from the repo root, one PART per call -- a call is limited to 20 minutes):
#   a: bench line, rehearsals of the N > 1 bench path on the shared GPU (2 and 4 ranks, gloo on the host side, the peer-read exchange between the processes,
#      BLA_BENCH_STRICT=1), data-parallel step floor, convolution shapes, batched U-Net (+ its rocprofv3 kernel statistics)
#   b: GEMM sweep, elementwise bandwidth, the bench under rocprofv3 and the headline kernel's three counter passes (FETCH_SIZE / WRITE_SIZE / MFMA busy;
#      the program itself after `--`)
#   c: per-workload kernel statistics + counter passes (tools/profile_r02.sh targets)
# usage: bash tools/measure_r03.sh OUTDIR a|b|c
out=${1:-gpurun_out/r03m}; part=${2:-a}; mkdir -p $out
export TMPDIR=/tmp
if [ $part = a ]; then
  python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
  for n in 2 4; do      # (the one-GPU box lets at most 6 processes use the card, the launcher counts as one or two: 8 ranks of the same command cannot be rehearsed there)
    BLA_BENCH_SHARE_GPU=1 BLA_BENCH_BACKEND=gloo BLA_BENCH_EXCHANGE=direct BLA_BENCH_STRICT=1 BLA_DP_MAX_BLOCKS=48 GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 \
      timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29517 + n)) \
      bench.py --gpus $n --steps 10 --warmup 5 --mnist-steps 100 > $out/bench_share$n.json 2> $out/bench_share$n.err; echo "bench share$n rc=$?"
  done
  python tools/dp_step_floor.py > $out/dp_step_floor.txt 2>&1; echo "dp floor rc=$?"
  python tools/conv_bench.py > $out/conv_bench.txt 2>&1; echo "conv bench rc=$?"
  python tools/unet_batch_bench.py 1 16 64 128 > $out/unet_batch.txt 2>&1; echo "unet batch rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/unet64_prof -- python3 tools/unet_batch_bench.py 64 > $out/unet64_prof.txt 2> $out/unet64_prof.err; echo "unet prof rc=$?"
  rm -f $out/unet64_prof/*/*_kernel_trace.csv      # the per-dispatch trace is tens of MB; the statistics are what is kept
elif [ $part = b ]; then
  python tools/gemm_sweep.py --sizes 1024,1536,2048,3072,4096,5120,6144,8192 --configs=-1 --layouts nn,nt --rounds 3 --iters 40 > $out/gemm_sweep.txt 2>&1; echo "sweep rc=$?"
  python tools/ew_bench.py 8192 > $out/ew_bench.txt 2>&1; echo "ew rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err; echo "bench prof rc=$?"
  rm -f $out/bench_prof/*/*_kernel_trace.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/gemm4096_$c -- python3 bench.py --no-cpu-baseline --mnist-steps 0 --conv-steps 0 --steps 10 --warmup 5 > $out/gemm4096_$c.json 2> $out/gemm4096_$c.err; echo "pmc $c rc=$?"
  done
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/gemm4096_mfma -- python3 bench.py --no-cpu-baseline --mnist-steps 0 --conv-steps 0 --steps 10 --warmup 5 > $out/gemm4096_mfma.json 2> $out/gemm4096_mfma.err; echo "pmc mfma rc=$?"
  python3 tools/gemm4096_traffic.py $out > $out/r03_gemm4096_traffic.json
else
  for shape in "4096 4096 1536" "4096 4096 4096"; do      # the headline kernel's matrix-pipe counter below its 2^31 saturation, and at the headline size
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/busy_${shape// /x} -- python3 tools/gemm_rect.py $shape 10 > $out/busy_${shape// /x}.txt 2>&1
    python3 tools/mfma_busy_summary.py $out/busy_${shape// /x} 100 >> $out/mfma_busy.txt 2>&1
  done
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/busy_conv128 -- python3 tools/profile_targets.py conv128 5 > $out/busy_conv128.txt 2>&1
  python3 tools/mfma_busy_summary.py $out/busy_conv128 >> $out/mfma_busy.txt 2>&1; echo "busy rc=$?"
  bash tools/profile_r02.sh $out/prof conv128 conv256 conv8 convs2 mnist mnist_dp softmax_cols add > $out/prof.log 2>&1; echo "profiles rc=$?"
  rm -f $out/prof/*/*/*/*_kernel_trace.csv
fi
echo done
