#!/bin/bash
# round 4 GPU lease wrapper: scripts/r4/call.sh <tag> '<command>'  - runs the command on the GPU box with stdout+stderr under gpurun_out/r4/<tag>.log
tag=$1; shift
mkdir -p gpurun_out/r4
( eval "$@" ) > gpurun_out/r4/$tag.log 2>&1
echo "exit $? : $tag"; tail -5 gpurun_out/r4/$tag.log
