#!/bin/bash
# End-to-end rate of the spotfinder driver on a GPU box: Eiger-16M frames as bitshuffle-LZ4 chunks in an
# Eiger-stream directory (32 distinct frames, 4096 directory entries), decoded on the GPU vs on the CPU.
#   gpurun -- 'bash tools/e2e_cli.sh'
set -e
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/tmp/e2e; rm -rf $T; mkdir -p $T
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm
cd $T/shm
N=4096
for i in $(seq 32 $((N-1))); do ln -s image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
sed -i "s/\"nimages\": 32/\"nimages\": $N/" start_1
cd $T
for cfg in "2 16" "4 16" "8 16" "8 32" "12 16"; do set -- $cfg; echo "gpu-decode threads $1 batch $2: $($B/spotfinder $T/shm --threads $1 --batch $2 2>&1 | grep -E 'images in')"; done
echo "cpu-decode threads 16 batch 4: $($B/spotfinder $T/shm --threads 16 --batch 4 --cpu-decode --images 512 2>&1 | grep -E 'images in')"
