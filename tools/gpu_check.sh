#!/bin/bash
# What a round's GPU check runs on the box, with stderr kept (the runtime's own fault / terminate message
# goes there):   gpurun -- 'bash tools/gpu_check.sh <tag> [bench args]'
tag=${1:-check}; shift
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -q ${PYTEST_X--x} > $out/gputests.log 2>&1; rc=$?; tail -4 $out/gputests.log
[ $rc -ne 0 ] && exit $rc
python bench.py "$@" > $out/bench.json 2> $out/bench.err; rc=$?; cat $out/bench.json; tail -2 $out/bench.err
exit $rc
