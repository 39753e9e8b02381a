#!/bin/bash
# two ranks on one GPU, mailboxes mapped through file descriptors: full stderr of both
cd $GRAFT_REPO_ROOT
export PISO_PEER_MAP=${1:-fd} HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1
python tests/slab_worker.py 0 2 29517 1024 1024 0 1 > /tmp/w0.out 2> /tmp/w0.err &
python tests/slab_worker.py 1 2 29517 1024 1024 0 1 > /tmp/w1.out 2> /tmp/w1.err
wait
for f in /tmp/w0.out /tmp/w0.err /tmp/w1.out /tmp/w1.err; do echo "== $f"; tail -c 2500 $f; done
