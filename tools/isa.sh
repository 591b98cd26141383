#!/bin/bash
# usage: tools/isa.sh <file.hip> <mangled-prefix> -> /tmp/isa_func.s
set -e
cd /root/repo/snickery_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../include -S --cuda-device-only "$1" -o /tmp/isa_all.s 2>&1 | grep -v "hip-link" || true
awk -v pat="^$2.*:" '$0 ~ pat {f=1} f{print} /^.Lfunc_end/{if(f){exit}}' /tmp/isa_all.s > /tmp/isa_func.s
wc -l /tmp/isa_func.s
