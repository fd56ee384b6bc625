#!/bin/bash
# Where the driver's time goes end to end (GPU box): 4096 distinct chunk files in /dev/shm, second pass over them, -v.
#   gpurun -- 'bash tools/e2e_diag.sh [images] ["flags;flags;..."]'
N=${1:-4096}
IFS=';' read -ra sets <<< "${2:---batch 16;--batch 8;--batch 32}"
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/dev/shm/ffs_diag_$$; rm -rf $T; mkdir -p $T
trap "rm -rf $T" EXIT
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
cd $T/shm
for i in $(seq 32 $((N-1))); do cp image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
sed -i "s/\"nimages\": 32/\"nimages\": $N/" start_1
cd $T
$B/spotfinder $T/shm --threads 16 --images $N > /dev/null 2>&1      # first pass over fresh files: not what is measured here
for rep in 1 2; do
for f in "${sets[@]}"; do
  echo "== $f (rep $rep)"
  $B/spotfinder $T/shm --threads 16 --images $N -v $f 2>&1 | grep -E "images in|collector|^Thread +[0-9]+: [0-9]+ chunks|assembly .* ready|Workers joined|submitted" | sed -e 's/\x1b\[[0-9;]*m//g' | awk '/chunks read/{c++; if (c<=3) print; next} {print}'
done
done
