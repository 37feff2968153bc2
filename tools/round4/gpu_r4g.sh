#!/bin/bash
T=gpurun_out/r4g; mkdir -p $T; export TMPDIR=/tmp
./probes/vmem_width 2>&1 | grep -v amdgpu.ids | tee $T/vmem_width.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $T/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $T/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4g/bench.json"))
print({k: d[k] for k in ("value", "value_packed", "ms_per_step", "encode_ms", "decode_ms", "whole_path_hbm_frac", "whole_path_hbm_frac_packed", "kernels_id")})
print(d["roofline"]); print(d["cpu_baseline"])
for k in d.get("data_kinds", []): print(k["data_kind"], k["value"], k["compression_ratio"], k["encode_ms"], k["decode_ms"], k["roofline"]["encode"]["frac"], k["roofline"]["decode"]["frac"])
for k in d.get("other_configs", []): print(k["config"][:60], k["value"], k.get("compression_ratio"), k["encode_ms"], k["decode_ms"])
PY
