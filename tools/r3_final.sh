#!/bin/bash
# the round's bench record + rocprofv3 kernel stats of the same command (raw traces deleted on the box: only the stats CSV comes back)
set -o pipefail
P=$PWD; O=$P/gpurun_out/${1:-r3final}; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $P/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/stats.log 2>&1
echo "stats rc=$?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf /tmp/prof_stats
head -n 12 $O/kernel_stats.csv | cut -c1-200
exit 0
