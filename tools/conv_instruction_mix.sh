out=$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/c_$i -- python3 tools/profile_targets.py conv256 5 > $out/c_$i.txt 2>&1 || echo "set $i failed"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/d_$i -- python3 tools/profile_targets.py conv128 5 > $out/d_$i.txt 2>&1 || echo "set $i failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    tag = f.split("/")[-4][0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_f32_glds" not in n: continue
        key = tag + " " + (n.split("<")[1].split(">")[0] if "<" in n else n)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v); print(f"   {c:28s} median {v[len(v)//2]:16.0f}  n {len(v)}")
PY
