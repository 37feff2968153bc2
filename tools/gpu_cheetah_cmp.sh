#!/bin/bash
# passes (variant 0) against the one-wave decoder (variant 128) per chunk size
T=gpurun_out/${1:-r3i}; mkdir -p $T; export TMPDIR=/tmp
for c in ${CHUNKS:-65536 131072 262144 524288 1048576}; do for v in 0 128; do
  timeout 600 python bench.py --algo cheetah --data prose --size 100000000 --chunk $c --variant $v --steps 3 --warmup 1 --no-cpu --no-sweep --no-extra > $T/b_${c}_$v.json 2> $T/b_${c}_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$T/b_${c}_$v.json")); print("chunk $c variant $v: decode", d["kernel_ms"]["cheetah_decode_chunks"], "encode", d["kernel_ms"]["cheetah_encode_chunks"], "ratio", d["compression_ratio"])
except Exception as ex: print("$c $v failed", ex, open("$T/b_${c}_$v.err").read()[-300:])
PY
done; done
