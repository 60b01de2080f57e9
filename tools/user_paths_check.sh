#!/bin/bash
# The commands a user of the reference types (reference README.md:10-21, initializer.py:128-184), on whatever machine
# this runs on: every mode / strategy through the root initializer.py, the framed-TCP transport, the three roles
# started by hand, one stand-alone entry point. Prints the tail of each run and its exit code.
# usage: bash tools/user_paths_check.sh
set -o pipefail
export DTF_SYNTH_MNIST_TRAIN=3000 DTF_SYNTH_MNIST_TEST=500
run() { echo "== $*"; timeout -k 10 200 "$@" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3; echo "rc=${PIPESTATUS[0]}"; }
run python initializer.py -m c -cs sync -n 3 -b 100
run python initializer.py -m c -cs async -n 2 -b 100 -ca true
run python initializer.py -m d -ds keras -n 2 -b 100
run python initializer.py -m d -ds graph -n 3 -d 2 -b 100
run python initializer.py -m d -ds custom -n 2 -b 100
run python initializer.py -m c -cs sync -n 2 -b 100 --transport tcp
# roles started by hand: one server, two workers
( timeout -k 10 200 python initializer.py -m c -cs sync -n 2 -b 100 -tt server > /tmp/server.log 2>&1; echo "server rc=$?" >> /tmp/server.log ) &
sleep 2
( timeout -k 10 200 python initializer.py -m c -cs sync -n 2 -b 100 -tt worker -ti 0 > /tmp/w0.log 2>&1; echo "w0 rc=$?" >> /tmp/w0.log ) &
timeout -k 10 200 python initializer.py -m c -cs sync -n 2 -b 100 -tt worker -ti 1 > /tmp/w1.log 2>&1; echo "w1 rc=$?" >> /tmp/w1.log
wait
echo "== roles by hand"; tail -3 /tmp/server.log; tail -2 /tmp/w0.log; tail -2 /tmp/w1.log
run python -m distributed_tensorflow_amd.cli.standalone keras -n 1 -b 100 -i 0 -w 127.0.0.1:6000 --max-steps 10
