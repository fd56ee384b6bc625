#!/bin/bash
# Why do the driver's workers read tmpfs at half the rate of tools/ubench/shm_read?  CPU / NUMA layout of the box, the bare
# reader under the same affinity, and the driver with --read-only (reads into the pinned staging buffers, submits nothing).
B=$GRAFT_REPO_ROOT/fast-feedback-service_amd/bin; T=/dev/shm/ffs_rd; rm -rf $T; mkdir -p $T
echo "nproc $(nproc); affinity $(taskset -p $$ 2>/dev/null | sed 's/.*: //'); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist) free $(grep MemFree $n/meminfo | awk '{print $4}') kB"; done
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "fast-feedback-service_amd", "python"))
import ffs_amd
from ffs_amd import api
print("GPU 0 NUMA node:", api.device_numa_node(0))
PY
$B/ffs_hosttool mkshm synth:eiger16m:32 $T/shm > /dev/null
N=${N:-1000}
cd $T/shm; for i in $(seq 32 $((N-1))); do cp image_$(printf %06d $((i%32)))_2 image_$(printf %06d $i)_2; done
sed -i "s/\"nimages\": 32/\"nimages\": $N/" start_1
cd $T
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
run() { local t0="$(thr)"; $B/spotfinder $T/shm --threads 8 --batch 4 -v "$@" > $T/out.txt 2> $T/err.txt
  echo "== 8 x 4 $* ${FFS_SHM_PLAIN_READ:+(plain read)}: $(grep -E 'images in' $T/out.txt | sed 's/\x1b\[[0-9;]*m//g')"; grep "batches; reading" $T/out.txt | sed -n '1,2p'; grep -E "CPU time|Workers joined|3D finish|3D analysis in all" $T/out.txt; echo "   cpu.stat before: $t0 after: $(thr)"; }
run
for k in 1 2 3; do
  export FFS_SHM_PLAIN_READ=1; run
  unset FFS_SHM_PLAIN_READ; run
done
run --threads 12 --all-threads
run --threads 16 --all-threads
ps -eLo pid,tid,pcpu,comm 2>/dev/null | sort -k3 -n -r | sed -n '1,5p'
[ -z "$SKIP_UBENCH" ] && $GRAFT_REPO_ROOT/tools/ubench/shm_read $T/shm 1000 | grep -E "registered +(4|8) threads|DMA +(4|8) threads"
rm -rf $T
