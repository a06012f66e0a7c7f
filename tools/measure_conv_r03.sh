#!/bin/bash
# Round-3 convolution measurements on the GPU box (run from the repo root through gpurun): parity first, then the batch-64 shapes on the padded-copy
# kernels (BLA_CONV_UNPADDED=0) and straight from the image (default), each with 128- and 256-wide forward tiles.
set -o pipefail
out=gpurun_out/${1:-r03_conv}
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_resnet.py tests/test_unet_batched.py -x -q > $out/tests.log 2>&1
rc=$?; echo "tests rc $rc" >> $out/tests.log; tail -4 $out/tests.log
[ $rc -eq 0 ] || exit $rc
for cfg in "UNPADDED=0 BN=128" "UNPADDED=1 BN=128" "UNPADDED=1 BN=256"; do
  set -- $cfg
  env BLA_CONV_${1} BLA_CONV_${2} CONV_BENCH_BATCH_ONLY=1 timeout -k 10 300 python tools/conv_bench.py > $out/conv_${1}_${2}.log 2>&1 || exit 1
  echo "== $cfg"; grep x64 $out/conv_${1}_${2}.log
done
timeout -k 10 300 python tools/unet_batch_bench.py > $out/unet_batch.log 2>&1; tail -6 $out/unet_batch.log
