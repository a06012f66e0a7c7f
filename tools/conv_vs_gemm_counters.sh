#!/bin/bash
# counters of the convolution gather kernels beside the plain half-slab GEMM on the same product (128 x 65536 x 1152)
out=$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/conv_$i -- python3 tools/profile_targets.py conv128 5 > $out/conv_$i.txt 2>&1 || echo "conv set $i failed"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/gemm_$i -- python3 tools/gemm_rect.py 128 65536 1152 5 > $out/gemm_$i.txt 2>&1 || echo "gemm set $i failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_f32_glds" not in n: continue
        key = n.split("<")[1].split(">")[0] if "<" in n else n
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v); print(f"   {c:28s} median {v[len(v)//2]:16.0f}  n {len(v)}")
PY
