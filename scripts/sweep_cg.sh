#!/bin/bash
for mb in 1024 2048; do for v in 2; do for r in 2 4 8; do echo "MAXBLOCKS=$mb V=$v RPW=$r"; PISO_CG_MAXBLOCKS=$mb PISO_CG_V=$v PISO_CG_RPW=$r python scripts/bench_cg.py 2048 2>&1 | grep grid; done; done; done
