#!/bin/bash
# The drop-in binary against another build of it (tools/ab/spotfinder_prev), same box, same files, alternating.
#   gpurun -- 'bash tools/e2e_ab.sh'
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; P=$GRAFT_REPO_ROOT/tools/ab/spotfinder_prev
T=/dev/shm/ffs_ab_$$; rm -rf $T; mkdir -p $T; trap "rm -rf $T" EXIT
N=4096
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
cd $T/shm
for i in $(seq 32 $((N-1))); do cp image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
sed -i "s/\"nimages\": 32/\"nimages\": $N/" start_1
cd $T
$B/spotfinder $T/shm --threads 16 --images $N > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 1000 4096; do
    for exe in $P $B/spotfinder; do
      [ -x $exe ] || continue
      # (as the service runs it: JSON lines into an inherited pipe, stdout into another)
      echo "$(basename $(dirname $exe))/$(basename $exe) $n: $($exe $T/shm --threads 16 --images $n --pipe_fd 3 3> >(cat > /dev/null) 2>&1 | grep -E 'images in' | sed -e 's/\x1b\[[0-9;]*m//g')"
    done
  done
done
