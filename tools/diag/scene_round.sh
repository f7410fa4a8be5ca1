#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 tools/diag/scene_form_check.py 2>&1 | grep "B=" | head -4
echo "== default"; python3 tools/diag/agg_parts_c4.py 2>&1 | grep " us" | head -3
for V in "$@"; do echo "== $V"; GROUPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/$V python3 tools/diag/agg_parts_c4.py 2>&1 | grep " us" | head -3; done
