mkdir -p gpurun_out/r03l && export TMPDIR=/tmp
for w in 1 0; do
BLA_CONV_WINDOW=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03l/w$w -- python3 tools/conv_k_sweep.py > gpurun_out/r03l/w$w.txt 2> gpurun_out/r03l/w$w.err || exit 1
cat gpurun_out/r03l/w$w.txt
done
