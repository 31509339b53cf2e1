#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r03 6ffe022 > gpurun_out/r03_profile.log 2>&1; tail -3 gpurun_out/r03_profile.log
bash tools/profile_round.sh r03_b65536 6ffe022 --batch 65536 > gpurun_out/r03_b65536_profile.log 2>&1; tail -3 gpurun_out/r03_b65536_profile.log
bash tools/profile_round.sh r03_bf16 6ffe022 --dtype bf16 > gpurun_out/r03_bf16_profile.log 2>&1; tail -3 gpurun_out/r03_bf16_profile.log
