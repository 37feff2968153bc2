#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5g: the whole GPU suite on the tree's library, then the bench line (paged container as the headline)
T=gpurun_out/r5g; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee $T/suite.txt
timeout 600 python bench.py > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"; tail -3 $T/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5g/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step", "compression_ratio", "encoded_bytes", "container_form", "value_packed", "value_slotted", "encode_ms", "decode_ms", "kernel_ms")})
print(r["roofline"])
print({k: r["cpu_baseline"][k] for k in ("value", "encode_MBps", "decode_MBps")}, r["cpu_baseline"]["all_cores"])
for o in r.get("data_kinds", []) + r.get("other_configs", []):
    print(o.get("config", "")[:60], o.get("value"), o.get("encode_ms"), o.get("decode_ms"), o.get("compression_ratio"), (o.get("cpu_baseline") or {}).get("all_cores"))
PY
