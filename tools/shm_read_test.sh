#!/bin/bash
# tmpfs -> worker-thread read rates on a GPU box (tools/ubench/shm_read.cc), then the driver with and without NUMA pinning
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/dev/shm/ffs_rd; rm -rf $T; mkdir -p $T
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
cd $T/shm; for i in $(seq 32 999); do cp image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
sed -i "s/\"nimages\": 32/\"nimages\": 1000/" start_1
$GRAFT_REPO_ROOT/tools/ubench/shm_read $T/shm 1000 | grep -E "read\(\)|DMA"
cd $T
for extra in "" "--no-numa-pinning"; do
  $B/spotfinder $T/shm --threads 8 --batch 4 -v $extra > $T/out.txt 2> $T/err.txt
  echo "== 8 x 4 $extra: $(grep -E 'images in' $T/out.txt | sed 's/\x1b\[[0-9;]*m//g')"; grep "batches; reading" $T/out.txt | sed -n '1,2p'
done
rm -rf $T
