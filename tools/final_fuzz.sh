#!/bin/bash
# round-4 final sweeps on the committed kernels (fresh seeds 7xxxxxx); the worker form of the tile pass forced in a second scenes / near leg
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
G='MISMATCH\|ERROR\|cases,\|soups,\|bad,\|Traceback'
( echo "== scenes"; timeout 1200 python tools/fuzz_parity.py 16000 7000000 | grep "$G"
  echo "== scenes, tile pass inside the shading launch forced"; DDX_BIG_INLINE=1 timeout 900 python tools/fuzz_parity.py 8000 7050000 | grep "$G"
  echo "== near"; FUZZ_NEAR=0.7 timeout 600 python tools/fuzz_parity.py 5000 7100000 | grep "$G"
  echo "== near, tile pass inside the shading launch forced"; DDX_BIG_INLINE=1 FUZZ_NEAR=0.7 timeout 600 python tools/fuzz_parity.py 4000 7150000 | grep "$G"
  echo "== big"; FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py 250 7200000 | grep "$G"
  echo "== far"; FUZZ_FAR=1 timeout 400 python tools/fuzz_parity.py 2500 7300000 | grep "$G"
  echo "== soups"; FUZZ_SOUPS=1 timeout 900 python tools/fuzz_parity.py 15000 7400000 | grep "$G"
  echo "== engine soups"; FUZZ_ENGINE_SOUPS=1 timeout 600 python tools/fuzz_parity.py 8000 7500000 | grep "$G"
  echo "== engine soups, tile pass inside the shading launch forced"; DDX_BIG_INLINE=1 FUZZ_ENGINE_SOUPS=1 timeout 600 python tools/fuzz_parity.py 5000 7550000 | grep "$G"
  echo "== state"; FUZZ_STATE=1 timeout 600 python tools/fuzz_parity.py 1500 7600000 | grep "$G"
  echo "== state, 32-64 hypotheses, every run of 2+ iterations as two chains against the one-chain run"; DDX_TWO_MIN=2 DDX_BIG_INLINE=1 FUZZ_WIDE=1 FUZZ_STATE=1 timeout 900 python tools/fuzz_parity.py 600 7650000 | grep "$G"
  echo "== api"; FUZZ_API=1 timeout 600 python tools/fuzz_parity.py 1200 7700000 | grep "$G"
  echo "== ops"; FUZZ_OPS=1 timeout 600 python tools/fuzz_parity.py 8000 7800000 | grep "$G" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c_fuzz_final.log
