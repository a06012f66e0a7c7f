#!/bin/bash
# rocprofv3 evidence for the kernels VERDICT r1 asked about: per target one --kernel-trace --stats run and two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE; the program itself after `--`).  usage (on the GPU box): bash tools/profile_r02.sh OUTDIR target...
set -e
out=$1; shift
export TMPDIR=/tmp
for t in "$@"; do
  mkdir -p $out/$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$t/stats -- python3 tools/profile_targets.py $t 20 > $out/$t/stats.json 2> $out/$t/stats.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$t/fetch -- python3 tools/profile_targets.py $t 5 > $out/$t/fetch.json 2> $out/$t/fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/$t/write -- python3 tools/profile_targets.py $t 5 > $out/$t/write.json 2> $out/$t/write.err
  python3 tools/profile_summary.py $out $t > $out/$t.summary.json
  echo "profiled $t"
done
