#!/bin/bash
# Runs a command on the GPU box with its output kept: stdout + stderr go to gpurun_out/<tag>_<UTC time stamp>.log, and the exit
# code is the file's last line -- a failing run and the repeat that follows it can never share a file name, so the runtime's own
# message of an abort is still there afterwards.
#   gpurun -- 'bash tools/run_logged.sh <tag> <command> [args...]'
tag=$1; shift
mkdir -p gpurun_out
log=gpurun_out/${tag}_$(date -u +%Y%m%dT%H%M%SZ)_$$.log
echo "# $(date -u +%FT%TZ) $*" > "$log"
"$@" >> "$log" 2>&1
rc=$?
echo "# exit code $rc" >> "$log"
echo "$log: exit code $rc"
tail -n 15 "$log"
exit $rc
