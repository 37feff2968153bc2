#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>" [source dir]  ->  probes/variants/lib_<name>.so  (experiment builds for same-box A/B runs:
# tools/gpu_variants.py; git-ignored, shipped to the GPU box with the tree)
set -e
NAME=$1; FLAGS=$2; SRC=${3:-density_amd/csrc}
mkdir -p probes/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DDENSITY_HIP_DEBUG -DDENSITY_HIP_KERNELS_ID="\"variant-$NAME\"" $FLAGS -o probes/variants/lib_$NAME.so \
  $SRC/api.hip $( [ -f $SRC/api_stream.hip ] && echo $SRC/api_stream.hip $SRC/api_host.hip ) $SRC/chameleon.hip $SRC/rotor.hip $SRC/container.hip $SRC/serial_codec.hip $SRC/stream_parse.hip $SRC/exchange_stages.hip $SRC/decode_passes.hip $( [ -f $SRC/placement.hip ] && echo $SRC/placement.hip )
ls -la probes/variants/lib_$NAME.so
