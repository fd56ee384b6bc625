#!/bin/bash
# Where a short request's wall time goes OUTSIDE the binary's own timer: start-up and tear-down of bin/spotfinder, one context and two
# (--devices 0,0), -v stamps inside main and the wall clock around the process.   gpurun -- 'bash tools/run_logged.sh <tag> bash tools/cli_startup.sh [images]'
N=${1:-1000}
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/dev/shm/ffs_startup_$$; rm -rf $T; mkdir -p $T; trap "rm -rf $T" EXIT
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
cd $T/shm
for i in $(seq 32 $((N-1))); do ln -s image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
sed -i "s/\"nimages\": 32/\"nimages\": $N/" start_1
cd $T
for flags in "" "--devices 0,0" "" "--devices 0,0"; do
  t0=$(date +%s.%N)
  $B/spotfinder $T/shm --threads 16 -v $flags > $T/out.txt 2> $T/err.txt
  rc=$?
  t1=$(date +%s.%N)
  echo "== flags [$flags] rc $rc wall $(python3 -c "print(round($t1-$t0,3))") s; $(grep -E 'images in' $T/out.txt | sed 's/\x1b\[[0-9;]*m//g')"
  grep -E "^\[ *[0-9.]+ ms\]" $T/out.txt
  if [ -s $T/err.txt ]; then echo "stderr:"; head -5 $T/err.txt; fi
done
