#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for k in 2 1; do for s in 16 32 48 64; do echo -n "NEEDLE_SHORT_WGS=$k "; NEEDLE_SHORT_WGS=$k python scripts/short_rows_rate.py $s 2>&1 | grep containedIn; done; done; done
