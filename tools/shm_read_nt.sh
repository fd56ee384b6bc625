#!/bin/bash
# tmpfs -> registered memory beside the GPU's DMA: the kernel's copy against non-temporal copies (tools/ubench/shm_read.cc modes 5-7)
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/dev/shm/ffs_rd; rm -rf $T; mkdir -p $T
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
cd $T/shm; for i in $(seq 32 999); do cp image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
$GRAFT_REPO_ROOT/tools/ubench/shm_read $T/shm 1000 | grep -E "registered|DMA" | grep -E " (4|8|16) threads|DMA beside"
rm -rf $T
