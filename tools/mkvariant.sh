#!/bin/bash
# usage: bash tools/mkvariant.sh NAME [extra hipcc flags]  -> tools/ab_NAME.so: the library with the working tree's
# tmx_attn_kernel.hip (work-list + band workgroups) compiled with the extra flags, every other object as built by make
name=$1; shift
cd /root/repo/vllm_trianglemix_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-inline-asm -mllvm -amdgpu-sched-strategy=iterative-maxocc -I. "$@" -c tmx_attn_kernel.hip -o /tmp/attn_$name.o 2>&1 | grep -E "error" -A3
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/tools/ab_$name.so $(ls _build/*.o | grep -v "tmx_attn_kernel\|_tune\|_diag\|_exp") /tmp/attn_$name.o
