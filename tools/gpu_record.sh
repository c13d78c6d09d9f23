#!/bin/bash
# The round's bench record + the rocprofv3 kernel stats of the same command (raw traces deleted on the box):
#   bash tools/gpu_record.sh OUT [bench.py arguments ...]
# writes gpurun_out/OUT/bench.json (the JSON line), bench.err, kernel_stats.csv.  Exit code = the first failing step's.
set -o pipefail
P=$PWD; O=$P/gpurun_out/${1:-record}; mkdir -p "$O"; shift
python3 bench.py "$@" > "$O/bench.json" 2> "$O/bench.err"; rc=$?
echo "bench rc=$rc"; cut -c1-300 "$O/bench.json"
[ $rc -eq 0 ] || { tail -n 20 "$O/bench.err"; exit $rc; }
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$P/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-fp32-pass --no-sampling-pass "$@" > "$O/stats.log" 2>&1
rc=$?
echo "stats rc=$rc"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
rm -rf /tmp/prof_stats
head -n 12 "$O/kernel_stats.csv" | cut -c1-200
exit $rc
