#!/bin/bash
# kernel-trace-only profile of one tools/profile_targets.py workload: per-kernel median durations.  usage: prof_quick.sh OUTDIR target [ENV=VAL ...]
out=$1; t=$2; shift 2
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf $out/$t.q; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/$t.q -- python3 tools/profile_targets.py $t 10 > /dev/null 2> $out/$t.q.err
python3 tools/kernel_trace_summary.py $out/$t.q
