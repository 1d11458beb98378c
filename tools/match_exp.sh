#!/bin/bash
# development: kernel-trace stats of dif_match under the timing-experiment builds of the one-term filter kernel
# (lib/libdif_eN.so = match.hip with -DB1_EXP=N, built by tools/build_variant.sh; results of those builds are invalid, only the kernel durations count)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/match_exp
mkdir -p $out
export MATCH_OPTS=filter=2
for lib in libdif.so libdif_e1.so libdif_e2.so libdif_e3.so libdif_e4.so; do
  [ -f $GRAFT_REPO_ROOT/deep-insight-face_amd/lib/$lib ] || continue
  export DIF_LIB=$lib
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$lib -o s -- python3 $GRAFT_REPO_ROOT/tools/match_prof.py 1000000 512 20 > $out/$lib.log 2>&1 || exit 1
  f=$(find $out/$lib -name '*kernel_stats.csv' | head -1)
  echo "== $lib: $(grep 'per dif_match' $out/$lib.log)" >> $out/summary.txt
  [ -n "$f" ] && grep -E "match_|hi1_|split2|probe_eps" $f | awk -F, '{printf "   %-40s calls %s avg %.1f us\n", substr($1,1,40), $2, $4/1000}' >> $out/summary.txt
done
cat $out/summary.txt
