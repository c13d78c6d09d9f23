set -o pipefail
O=gpurun_out/${1:-sweep}; mkdir -p $O
for B in 16 64 128; do
  python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --no-sampling-pass > $O/b$B.json 2> $O/b$B.err || { tail -n 5 $O/b$B.err; exit 1; }
  echo "batch $B done: $(cut -c1-120 $O/b$B.json)"
done
python3 bench.py --workload scaled --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-pass --no-sampling-pass > $O/scaled_b8.json 2> $O/scaled_b8.err || { tail -n 5 $O/scaled_b8.err; exit 1; }
echo "scaled b8 done"
python3 bench.py --workload scaled --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --no-sampling-pass > $O/scaled_b64.json 2> $O/scaled_b64.err || { tail -n 5 $O/scaled_b64.err; exit 1; }
echo "scaled b64 done"
python3 bench.py --force-dp --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --no-sampling-pass 2> $O/dp1.err | tail -n 1 > $O/dp1.json || { tail -n 5 $O/dp1.err; exit 1; }
echo "dp1 rccl done: $(cut -c1-100 $O/dp1.json)"
