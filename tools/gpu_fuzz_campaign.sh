#!/bin/bash
# a long run of every fuzzer (beside the closing recipe's short ones): bash tools/gpu_fuzz_campaign.sh [scale]   -> gpurun_out/fuzz_campaign.txt
S=${1:-1}; export TMPDIR=/tmp; O=gpurun_out/fuzz_campaign.txt; : > $O
run() { echo "== $*" >> $O; timeout 1500 "$@" 2>&1 | grep -v amdgpu.ids | grep -E "DIVERGENCE|FAIL|divergences|failures|trials" | tail -40 >> $O; }
run python tools/gpu_fuzz_encode.py $((1500 * S)) 5
run python tools/gpu_fuzz_encode.py $((1500 * S)) 6
run python tools/gpu_fuzz_streams.py $((1500 * S))
run python tools/gpu_fuzz_passes.py $((1200 * S))
run python tools/gpu_fuzz_tail.py $((2500 * S)) 41
run python tools/gpu_fuzz_paged.py $((3000 * S)) 42
grep -c "DIVERGENCE\|FAIL" $O
