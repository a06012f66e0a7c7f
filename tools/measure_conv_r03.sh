#!/bin/bash
# Round-3 convolution measurements on the GPU box (run from the repo root through gpurun): parity first, then the batch-64 shapes on the padded-copy
# kernels (BLA_CONV_WINDOW=0) and with the image window in LDS (default), then the batched U-Net.
set -o pipefail
out=gpurun_out/${1:-r03_conv}
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_resnet.py tests/test_unet_batched.py -x -q > $out/tests.log 2>&1
rc=$?; echo "tests rc $rc" >> $out/tests.log; tail -4 $out/tests.log
[ $rc -eq 0 ] || exit $rc
for w in 0 1; do
  BLA_CONV_WINDOW=$w CONV_BENCH_BATCH_ONLY=1 timeout -k 10 300 python tools/conv_bench.py > $out/conv_window$w.log 2>&1 || exit 1
  echo "== BLA_CONV_WINDOW=$w"; grep x64 $out/conv_window$w.log
done
timeout -k 10 300 python tools/unet_batch_bench.py 64 > $out/unet_batch.log 2>&1; tail -3 $out/unet_batch.log
