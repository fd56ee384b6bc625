#!/bin/bash
# Where the drop-in binary's wall time goes on a GPU box: stream set-up against steady state, by threads x batch.
#   gpurun -- 'bash tools/cli_profile.sh [n_images] ["threads batch" ...]'
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/dev/shm/ffs_prof; rm -rf $T; mkdir -p $T
N=${1:-1000}; shift
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
cd $T/shm
if [ -n "$FFS_PROFILE_COPIES" ]; then   # distinct files, as a detector writes them (7.5 GB of tmpfs for 1000 frames)
  for i in $(seq 32 $((N-1))); do cp image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
else
  for i in $(seq 32 $((N-1))); do ln -s image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
fi
sed -i "s/\"nimages\": 32/\"nimages\": $N/" start_1
cd $T
[ $# -eq 0 ] && set -- "8 4" "6 4" "4 8" "12 2" "16 4"
for cfg in "$@"; do
  set -- $cfg
  $B/spotfinder $T/shm --threads $1 --batch $2 -v $3 > $T/out.txt 2> $T/err.txt
  echo "== threads $1 batch $2 $3: $(grep -E 'images in' $T/out.txt | sed 's/\x1b\[[0-9;]*m//g')"
  grep "streams ready" $T/out.txt | sort -t'(' -k2 -n | sed -n '1p;$p'
  grep "batches; reading" $T/out.txt | sed -n '1,3p'
done
rm -rf $T
