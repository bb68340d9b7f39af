#!/bin/bash
# One GPU-box session of round 4.  Usage (repo root on the GPU box):  bash tools/gpu_round4.sh <tag> [tests|bench|quick|sched|prof|full]
#   tests  pytest -m gpu + smoke     bench  the contract bench     quick  tests + bench     sched  K_sched counter passes (bench rows)
#   prof   rocprofv3 kernel trace + PMC passes of the bench command (tools/gpu_round3.sh prof)
set -u
TAG=${1:-r08}
MODE=${2:-quick}
has() { [[ "|$1|" == *"|$MODE|"* ]]; }
if has "tests|bench|quick"; then bash tools/gpu_round3.sh "$TAG" "$MODE"; fi
if has "full"; then bash tools/gpu_round3.sh "$TAG" quick; fi
if has "sched|full"; then
  bash tools/sched_counters.sh "$TAG" 2>&1 | tail -40
fi
if has "prof|full"; then bash tools/gpu_round3.sh "${TAG}" prof; fi
