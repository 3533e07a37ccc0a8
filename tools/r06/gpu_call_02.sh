#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c02; mkdir -p $O
timeout 300 tools/ubench/_build/valu_facts2.out > $O/valu_facts2.log 2>&1; echo "rc $?"
cat $O/valu_facts2.log
