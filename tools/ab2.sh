#!/bin/bash
# tools/ab2.sh "VAR1=a VAR2=b" "VAR1=c" ...: per-kernel times for each environment setting
for v in "$@"; do
  echo "== $v"
  env $v python tools/ktimes.py 2>&1 | grep -v amdgpu.ids || exit 1
done
