#!/bin/bash
# round 6: Cheetah's one-wave encoder two records per step — parity (every Cheetah / Lion suite, the encode fuzzer), then config 3
T=gpurun_out/r6g; mkdir -p $T; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py tests/test_gpu_patchwork.py tests/test_gpu_decode_passes.py tests/test_gpu_c_example.py -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 300 python tools/gpu_fuzz_encode.py > $T/fuzz_encode.log 2>&1; echo "fuzz rc=$?"; tail -2 $T/fuzz_encode.log
timeout 300 python bench.py --algo cheetah --data prose --size 100000000 --steps 8 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_cheetah.json 2> $T/bench_cheetah.err
python - <<PY
import json
try:
    d=json.load(open("$T/bench_cheetah.json")); print("cheetah:", d["value"], "MB/s; kernel_ms", d["kernel_ms"], "ratio", d["compression_ratio"])
except Exception as ex: print("failed", ex, open("$T/bench_cheetah.err").read()[-1500:])
PY
timeout 300 python benches/density.py 2>&1 | grep -A12 "cheetah" | head -14
