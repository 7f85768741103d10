#!/bin/bash
# usage: bash tools/mkvariant.sh NAME [extra hipcc flags]  -> tools/ab_NAME.so from the working tree's tmx_attention.hip
name=$1; shift
cd /root/repo/vllm_trianglemix_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-inline-asm -mllvm -amdgpu-sched-strategy=iterative-maxocc -I. "$@" -c tmx_attention.hip -o /tmp/attn_$name.o 2>&1 | grep -E "error" -A3
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/tools/ab_$name.so _build/tmx_api.o _build/tmx_mask.o _build/tmx_index.o /tmp/attn_$name.o _build/tmx_dense.o _build/tmx_decode.o _build/tmx_eager_f32.o _build/tmx_cache.o
