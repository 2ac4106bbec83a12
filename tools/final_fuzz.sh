#!/bin/bash
# tools/final_fuzz.sh <tag> [percent] -- the randomised sweeps on the committed kernels (fresh seeds 8xxxxxx); percent scales the case counts
# (100 = the full battery of round 4, ~1.5 h of GPU time; the round-5 session ran 30)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r6d}; P=${2:-30}
mkdir -p gpurun_out
n() { echo $(( $1 * P / 100 )); }
G='MISMATCH\|ERROR\|cases,\|soups,\|bad,\|Traceback'
( echo "== scenes"; timeout 1200 python tools/fuzz_parity.py $(n 16000) 8000000 | grep "$G"
  echo "== scenes, tile pass inside the shading launch forced"; DDX_BIG_INLINE=1 timeout 900 python tools/fuzz_parity.py $(n 8000) 8050000 | grep "$G"
  echo "== near"; FUZZ_NEAR=0.7 timeout 600 python tools/fuzz_parity.py $(n 5000) 8100000 | grep "$G"
  echo "== near, tile pass inside the shading launch forced"; DDX_BIG_INLINE=1 FUZZ_NEAR=0.7 timeout 600 python tools/fuzz_parity.py $(n 4000) 8150000 | grep "$G"
  echo "== big"; FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py $(n 250) 8200000 | grep "$G"
  echo "== far"; FUZZ_FAR=1 timeout 400 python tools/fuzz_parity.py $(n 2500) 8300000 | grep "$G"
  echo "== soups"; FUZZ_SOUPS=1 timeout 900 python tools/fuzz_parity.py $(n 15000) 8400000 | grep "$G"
  echo "== engine soups"; FUZZ_ENGINE_SOUPS=1 timeout 600 python tools/fuzz_parity.py $(n 8000) 8500000 | grep "$G"
  echo "== engine soups, tile pass inside the shading launch forced"; DDX_BIG_INLINE=1 FUZZ_ENGINE_SOUPS=1 timeout 600 python tools/fuzz_parity.py $(n 5000) 8550000 | grep "$G"
  echo "== state"; FUZZ_STATE=1 timeout 600 python tools/fuzz_parity.py $(n 1500) 8600000 | grep "$G"
  echo "== state, 32-64 hypotheses, every run of 2+ iterations as two chains against the one-chain run"; DDX_TWO_MIN=2 DDX_BIG_INLINE=1 FUZZ_WIDE=1 FUZZ_STATE=1 timeout 900 python tools/fuzz_parity.py $(n 600) 8650000 | grep "$G"
  echo "== api"; FUZZ_API=1 timeout 600 python tools/fuzz_parity.py $(n 1200) 8700000 | grep "$G"
  echo "== ops"; FUZZ_OPS=1 timeout 600 python tools/fuzz_parity.py $(n 8000) 8800000 | grep "$G" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_fuzz_final.log
