#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 4 8; do echo "CODA_ATTN_DIRECT=$v"; CODA_ATTN_DIRECT=$v timeout 300 python tools/bench_attn.py fp32 2>&1 | grep -E "L=  256|L=  512"; done
