#!/bin/bash
for p in 0 1; do echo "PERSIST=$p"; PISO_CG_PERSIST=$p python scripts/bench_cg.py 2048 1024 512 2>&1 | grep grid; done
for seg in 50 1000; do echo "SEGMENT=$seg"; PISO_CG_SEGMENT=$seg python scripts/bench_cg.py 2048 2>&1 | grep grid; done
