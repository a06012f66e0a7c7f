#!/bin/bash
# Round-2 measurement pass on one MI355X: bench line, rehearsal of the N > 1 bench path on the shared GPU, rocprofv3 summaries, tool outputs.
# usage (GPU box): bash tools/measure_r02.sh OUTDIR
out=${1:-gpurun_out/r02m}; mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
# two ranks on the one GPU (gloo for the host side, the direct exchange between the two processes): exercises the N > 1 code path of bench.py
BLA_BENCH_SHARE_GPU=1 BLA_BENCH_BACKEND=gloo BLA_BENCH_EXCHANGE=direct HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 5 --mnist-steps 100 > $out/bench_share2.json 2> $out/bench_share2.err; echo "bench share2 rc=$?"
python tools/dp_step_floor.py > $out/dp_step_floor.txt 2>&1
python tools/conv_bench.py > $out/conv_bench.txt 2>&1
python tools/unet_model_bench.py --oracle > $out/unet_model.txt 2>&1
python tools/unet_bench.py > $out/unet_blocks.txt 2>&1
python tools/unet_batch_bench.py 1 16 64 128 > $out/unet_batch.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/unet64_prof -- python3 tools/unet_batch_bench.py 64 > $out/unet64_prof.txt 2> $out/unet64_prof.err
python tools/ew_bench.py 8192 > $out/ew_bench.txt 2>&1
python tools/gemm_sweep.py --sizes 1024,2048,3072,4096,5120,6144,8192 --configs=-1 --layouts nn,nt,tn,tt --rounds 3 --iters 40 > $out/gemm_sweep.txt 2>&1
python tools/wsk_tile_compare.py > $out/wsk_tiles.txt 2>&1
python tools/c_trainer_e2e.py 60000 3 256 > $out/c_trainer_e2e.txt 2>&1
bash tools/profile_r02.sh $out/prof conv128 conv256 conv8 convs2 mnist mnist_dp softmax_cols transpose add colsum rowsum > $out/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
echo done
