#!/bin/bash
# one GPU call: node-form check, launch parts for the default library and variants, stamps
cd "$GRAFT_REPO_ROOT"
python3 tools/diag/node_form_check.py 2>&1 | grep "B=" 
echo "== default"; python3 tools/diag/agg_parts.py 2>&1 | grep " us"
for V in "$@"; do echo "== $V"; GROUPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/$V python3 tools/diag/agg_parts.py 2>&1 | grep " us"; done
python3 tools/diag/stamps_fwd.py stamps.bin 2>&1 | grep -A5 "agg_mlp_grouped"
