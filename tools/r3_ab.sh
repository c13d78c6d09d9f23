#!/bin/bash
# A/B of several stamps libraries on one box (graph replay): bash tools/r3_ab.sh OUTDIR tag1 tag2 ...   (tag "new" = libparler_tts_hip_stamps.so)
set -o pipefail
P=$PWD; O=$P/gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
for tag in "$@"; do
  lib=libparler_tts_hip_stamps_$tag.so; [ "$tag" = new ] && lib=libparler_tts_hip_stamps.so
  PTTS_HIP_LIB=$P/parler_tts_amd/csrc/$lib PTTS_STAMPS_OUT=$O/st_$tag.npy python3 tools/pmc_attn.py 40 24 1250 > $O/st_$tag.log 2>&1
  python3 tools/stamps_report.py $O/st_$tag.npy > $O/st_${tag}_$rep.txt 2>&1
  echo "== $tag ($rep)"; tail -n 10 $O/st_${tag}_$rep.txt | cut -c1-170
done
done
rm -f $O/*.npy
exit 0
